"""Generates tests/golden/lbs_golden.npz by IMPORTING the reference's own Python (read-only at
/root/reference) in the build container.  The reference source never ships; only these vectors do.

    python tests/golden/gen_lbs_golden.py

G1 rotations.py  : matrix_to_quaternion / rotation_6d_to_matrix / quaternion_multiply / quaternion_to_matrix
G2 body_model/smpl.py : lbs + batch_rigid_transform on a seeded synthetic SMPL-shaped model, 3 poses
G3 body_model/lbs.py  : lbs_extra for J in {24, 52}
G4 composite of sings_hybrid.py:398-428 built from the imported pieces (with / without ext_tfs)
G5 camera matrices of the shipped kit (graphics.py:65-85 is cv2-gated -> values from cameras.npz + our helper
   are pinned against numbers computed here with torch ops identical to Customdataset.py:128-134)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from sings.rec.utils.geometry import rotations as rot            # noqa: E402  (torch only)
from sings.rec.utils.body_model import smpl as vsmpl             # noqa: E402  (vendored SMPL math)

stub = types.ModuleType("smplx.lbs")
for fn in ("batch_rodrigues", "blend_shapes", "vertices2joints", "batch_rigid_transform"):
    setattr(stub, fn, getattr(vsmpl, fn))
sys.modules["smplx"] = types.ModuleType("smplx")
sys.modules["smplx.lbs"] = stub
from sings.rec.utils.body_model import lbs as rlbs               # noqa: E402  (lbs_extra)

from oracle.lbs_oracle import synthetic_body_model               # noqa: E402  (inputs only)

out = {}
rs = np.random.RandomState(1234)
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))

# ---- G1
R_ortho = rot.random_rotations(128).numpy() if hasattr(rot, "random_rotations") else None
mats = np.concatenate([R_ortho, R_ortho + 0.05 * rs.normal(size=(128, 3, 3)).astype(np.float32)], 0).astype(np.float32)
mats[5] = np.diag([1, -1, -1]); mats[6] = np.diag([-1, 1, -1]); mats[7] = np.diag([-1, -1, 1]); mats[8] = np.eye(3)
out["g1_mats"] = mats
out["g1_m2q"] = rot.matrix_to_quaternion(T(mats)).numpy()
d6 = rs.normal(size=(256, 6)).astype(np.float32)
out["g1_d6"] = d6; out["g1_d6_to_mat"] = rot.rotation_6d_to_matrix(T(d6)).numpy()
qa = rs.normal(size=(256, 4)).astype(np.float32); qb = rs.normal(size=(256, 4)).astype(np.float32)
out["g1_qa"] = qa; out["g1_qb"] = qb
out["g1_qmul"] = rot.quaternion_multiply(T(qa), T(qb)).numpy()
out["g1_q2m"] = rot.quaternion_to_matrix(T(qa)).numpy()

# ---- G2
bm = synthetic_body_model(seed=0)
betas = rs.normal(0, 1, (1, 10)).astype(np.float32)
amass = np.load(os.path.join(REF, "data/animation/AMASS/SFU/0008/0008_Walking002_poses.npz"))
AMASS_SMPLH_TO_SMPL_JOINTS = np.arange(0, 156).reshape((-1, 3))[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                                                   16, 17, 18, 19, 20, 21, 22, 37]].reshape(-1)
pose_amass = amass["poses"][100, AMASS_SMPLH_TO_SMPL_JOINTS].astype(np.float32)
da = np.zeros(72, np.float32); da[3 + 2] = 1.0; da[3 + 5] = -1.0          # get_predefined_pose('da_pose') on body_pose
poses = np.stack([np.zeros(72, np.float32), da, pose_amass])
out["g2_betas"] = betas; out["g2_poses"] = poses
parents = torch.from_numpy(bm["parents"])
posedirs0 = torch.zeros(207, bm["v_template"].shape[0] * 3)   # ignored by the vendored lbs (v_posed = v_shaped, smpl.py:339)
for i, p in enumerate(poses):
    verts, Jt = vsmpl.lbs(T(betas), T(p[None]), T(bm["v_template"][None]), T(bm["shapedirs"]), posedirs0, T(bm["J_regressor"]),
                          parents, T(bm["lbs_weights"]))
    v_shaped = T(bm["v_template"][None]) + vsmpl.blend_shapes(T(betas), T(bm["shapedirs"]))
    Jrest = vsmpl.vertices2joints(T(bm["J_regressor"]), v_shaped)
    Rm = vsmpl.batch_rodrigues(T(p).view(-1, 3)).view(1, -1, 3, 3)
    Jt2, A = vsmpl.batch_rigid_transform(Rm, Jrest, parents)
    out[f"g2_verts_{i}"] = verts.numpy(); out[f"g2_J_{i}"] = Jt.numpy(); out[f"g2_A_{i}"] = A.numpy()

# ---- G3 / G4
for J in (24, 52):
    N = 1000
    A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1))
    for j in range(J):
        Rj = vsmpl.batch_rodrigues(T(rs.normal(0, 0.6, (1, 3)))).numpy()[0]
        A[j, :3, :3] = Rj; A[j, :3, 3] = rs.normal(0, 0.2, 3)
    w = rs.rand(N, J).astype(np.float32) ** 5; w /= w.sum(1, keepdims=True)
    v = rs.normal(0, 0.4, (N, 3)).astype(np.float32)
    verts, A_, T_, _, _ = rlbs.lbs_extra(T(A)[None], T(v)[None], None, T(w), torch.zeros(1, J * 3), disable_posedirs=True,
                                        pose2rot=True)
    out[f"g3_A_{J}"] = A; out[f"g3_w_{J}"] = w; out[f"g3_v_{J}"] = v
    out[f"g3_verts_{J}"] = verts.numpy()[0]; out[f"g3_T_{J}"] = T_.numpy()[0]
    # composite sings_hybrid.py:398-428 restated with the imported pieces
    Rc = rot.rotation_6d_to_matrix(T(rs.normal(size=(N, 6)))); scales = T(np.exp(rs.normal(-4, 0.3, (N, 3))))
    smpl_scale = T([1.07]); transl = T([-0.04, 0.09, 10.06])
    xyz = verts[0] * smpl_scale.unsqueeze(0); sc = scales * smpl_scale.unsqueeze(0)
    xyz = xyz + transl.unsqueeze(0)
    Rdef = T_[0][:, :3, :3] @ Rc
    q = rot.matrix_to_quaternion(Rdef)
    out[f"g4_Rc_{J}"] = Rc.numpy(); out[f"g4_scales_{J}"] = scales.numpy()
    out[f"g4_xyz_{J}"] = xyz.numpy(); out[f"g4_q_{J}"] = q.numpy(); out[f"g4_sc_{J}"] = sc.numpy()
    trans = T([0.3, -0.1, 0.5]); rotm = vsmpl.batch_rodrigues(T([[0.2, 0.9, -0.4]]))[0]; scl = T([1.3])
    xyz2 = (trans[..., None] + (scl[None] * (rotm @ xyz[..., None]))).squeeze(-1)
    sc2 = scl * sc
    q2 = rot.quaternion_multiply(rot.matrix_to_quaternion(rotm), q)
    out[f"g4_ext_trans_{J}"] = trans.numpy(); out[f"g4_ext_rot_{J}"] = rotm.numpy(); out[f"g4_ext_scale_{J}"] = scl.numpy()
    out[f"g4_xyz_ext_{J}"] = xyz2.numpy(); out[f"g4_q_ext_{J}"] = q2.numpy(); out[f"g4_sc_ext_{J}"] = sc2.numpy()

# ---- G5: camera of the shipped kit, computed with the torch expressions of Customdataset.py:102-134
cam = np.load(os.path.join(REF, "examples/training_kits/f_2/score_demo_video/cameras.npz"))
K, E = cam["intrinsic"], cam["extrinsic"]; W, H = int(cam["width"]), int(cam["height"])
import math
fovx = 2 * math.atan(W / (2 * K[0, 0])); fovy = 2 * math.atan(H / (2 * K[1, 1]))
tanHalfFovY = math.tan((fovy / 2)); tanHalfFovX = math.tan((fovx / 2))
top = tanHalfFovY * 0.01; bottom = -top; right = tanHalfFovX * 0.01; left = -right
P = torch.zeros(4, 4)
P[0, 0] = 2.0 * 0.01 / (right - left); P[1, 1] = 2.0 * 0.01 / (top - bottom)
P[0, 2] = (right + left) / (right - left); P[1, 2] = (top + bottom) / (top - bottom)
P[3, 2] = 1.0; P[2, 2] = 100.0 / (100.0 - 0.01); P[2, 3] = -(100.0 * 0.01) / (100.0 - 0.01)
wvt = torch.from_numpy(E.astype(np.float32)).transpose(0, 1)
full = wvt.unsqueeze(0).bmm(P.transpose(0, 1).unsqueeze(0)).squeeze(0)
out["g5_K"] = K; out["g5_E"] = E; out["g5_WH"] = np.array([W, H])
out["g5_fov"] = np.array([fovx, fovy]); out["g5_wvt"] = wvt.numpy(); out["g5_full"] = full.numpy()
out["g5_center"] = wvt.inverse()[3, :3].numpy()
# a few real poses of the kit (inputs for LBS tests / bench scenes)
po = np.load(os.path.join(REF, "examples/training_kits/f_2/score_demo_video/poses_optimized.npz"))
out["kit_global_orient"] = po["global_orient"][:4]; out["kit_body_pose"] = po["body_pose"][:4]
out["kit_transl"] = po["transl"][:4]
out["amass_poses_72"] = amass["poses"][::24][:120][:, AMASS_SMPLH_TO_SMPL_JOINTS].astype(np.float32)

dst = os.path.join(ROOT, "tests", "golden", "lbs_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")
