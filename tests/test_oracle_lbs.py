"""Pins oracle/lbs_oracle.py (and sings_amd/camera.py) against tests/golden/lbs_golden.npz: outputs of
the reference's own rotations.py / body_model/smpl.py / body_model/lbs.py imported in the build container
(generator: tests/golden/gen_lbs_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lbs_golden.npz"))
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))


def close(a, b, tol=1e-6):
    a = a.numpy() if torch.is_tensor(a) else a
    assert np.abs(a - b).max() <= tol, np.abs(a - b).max()


def test_g1_rotations():
    close(lo.matrix_to_quaternion(T(G["g1_mats"])), G["g1_m2q"], 0)          # same ops -> identical
    close(lo.rotation_6d_to_matrix(T(G["g1_d6"])), G["g1_d6_to_mat"], 1e-7)
    close(lo.quaternion_multiply(T(G["g1_qa"]), T(G["g1_qb"])), G["g1_qmul"], 0)
    close(lo.quaternion_to_matrix(T(G["g1_qa"])), G["g1_q2m"], 1e-6)
    n = np.linalg.norm(G["g1_m2q"][128:], axis=1)
    assert n.min() < 0.999 or n.max() > 1.001        # non-orthonormal input -> non-unit quaternion is expected


def test_g2_smpl_chain():
    bm = lo.synthetic_body_model(seed=0)
    parents = bm["parents"].tolist()
    for i in range(3):
        verts, Jt, A, _ = lo.smpl_lbs(T(G["g2_betas"]), T(G["g2_poses"][i][None]), T(bm["v_template"][None]),
                                      T(bm["shapedirs"]), T(bm["J_regressor"]), parents, T(bm["lbs_weights"]))
        close(A, G[f"g2_A_{i}"], 2e-6)
        close(Jt, G[f"g2_J_{i}"], 2e-6)
        close(verts, G[f"g2_verts_{i}"], 5e-6)
    # zero pose: every A is (numerically) the identity
    assert np.abs(G["g2_A_0"][0] - np.eye(4)).max() < 1e-5


@pytest.mark.parametrize("J", [24, 52])
def test_g3_lbs_extra(J):
    verts, Tm = lo.lbs_extra(T(G[f"g3_A_{J}"])[None], T(G[f"g3_v_{J}"])[None], T(G[f"g3_w_{J}"]))
    close(verts[0], G[f"g3_verts_{J}"], 2e-6)
    close(Tm[0], G[f"g3_T_{J}"], 2e-6)


@pytest.mark.parametrize("J", [24, 52])
def test_g4_deform_composite(J):
    args = (T(G[f"g3_v_{J}"]), T(G[f"g4_Rc_{J}"]), T(G[f"g4_scales_{J}"]), T(G[f"g3_w_{J}"]), T(G[f"g3_A_{J}"]))
    xyz, q, sc, _ = lo.deform_gaussians(*args, smpl_scale=T([1.07]), transl=T([-0.04, 0.09, 10.06]))
    close(xyz, G[f"g4_xyz_{J}"], 5e-6); close(q, G[f"g4_q_{J}"], 5e-6); close(sc, G[f"g4_sc_{J}"], 1e-8)
    ext = (T(G[f"g4_ext_trans_{J}"]), T(G[f"g4_ext_rot_{J}"]), T(G[f"g4_ext_scale_{J}"]))
    xyz, q, sc, _ = lo.deform_gaussians(*args, smpl_scale=T([1.07]), transl=T([-0.04, 0.09, 10.06]), ext_tfs=ext)
    close(xyz, G[f"g4_xyz_ext_{J}"], 1e-5); close(q, G[f"g4_q_ext_{J}"], 5e-6); close(sc, G[f"g4_sc_ext_{J}"], 1e-8)


def test_g5_camera_of_shipped_kit():
    from sings_amd.camera import make_camera
    K, E = G["g5_K"], G["g5_E"]; W, H = [int(x) for x in G["g5_WH"]]
    cam = make_camera(E, K[0, 0], K[1, 1], K[0, 2], K[1, 2], W, H)
    assert abs(cam["fovx"] - G["g5_fov"][0]) < 1e-12 and abs(cam["fovy"] - G["g5_fov"][1]) < 1e-12
    close(cam["world_view_transform"], G["g5_wvt"], 0)
    close(cam["full_proj_transform"], G["g5_full"], 1e-6)
    close(cam["camera_center"], G["g5_center"], 1e-7)
    assert cam["image_width"] == 512 and cam["image_height"] == 896


def test_body_joint_transforms_match_reference_chain():
    """sings_amd/body.py (the product-side producer of A) vs the reference's batch_rigid_transform (golden G2)."""
    from sings_amd.body import joint_transforms
    bm = lo.synthetic_body_model(seed=0)
    v_shaped = T(bm["v_template"]) + torch.einsum("l,mkl->mk", T(G["g2_betas"][0]), T(bm["shapedirs"]))
    J_rest = torch.einsum("ik,ji->jk", v_shaped, T(bm["J_regressor"]))
    for i in range(3):
        A = joint_transforms(T(G["g2_poses"][i]), J_rest, tuple(bm["parents"].tolist()))
        close(A, G[f"g2_A_{i}"][0], 3e-6)
    # differentiable w.r.t. the pose
    pose = T(G["g2_poses"][2]).clone().requires_grad_(True)
    joint_transforms(pose, J_rest).square().sum().backward()
    assert torch.isfinite(pose.grad).all() and pose.grad.abs().max() > 0


def test_posed_driver_batch_chain_and_amass_mapping():
    """sings_amd/posed.py: chunked A production equals the per-frame chain (hence golden G2); AMASS joint map."""
    from sings_amd.body import joint_transforms
    from sings_amd.posed import AMASS_SMPLH_TO_SMPL_JOINTS, amass_to_smpl_pose, joint_transforms_batch
    bm = lo.synthetic_body_model(seed=0)
    J_rest = torch.einsum("ik,ji->jk", T(bm["v_template"]), T(bm["J_regressor"]))
    poses = T(G["amass_poses_72"][:6])
    A = joint_transforms_batch(poses, J_rest, tuple(bm["parents"].tolist()))
    for f in range(6):
        close(A[f], joint_transforms(poses[f], J_rest, tuple(bm["parents"].tolist())).numpy(), 1e-6)
    assert AMASS_SMPLH_TO_SMPL_JOINTS.shape == (72,) and list(AMASS_SMPLH_TO_SMPL_JOINTS[-3:]) == [111, 112, 113]
    p156 = np.arange(2 * 156, dtype=np.float32).reshape(2, 156)
    assert np.array_equal(amass_to_smpl_pose(p156)[:, :69], p156[:, :69]) and np.array_equal(amass_to_smpl_pose(p156)[:, 69:], p156[:, 111:114])
