"""CPU: the ORACLE's rotation conversions (oracle/rotations_oracle.py -- what the HIP kernels of sings_amd/rotations.py are
checked against in tests/test_gpu_rotations.py) and the orbit / static cameras (sings_amd/camera.py) against golden
vectors produced by the reference's own functions (tests/golden/gen_rot_cam_golden.py)."""
import os

import numpy as np
import torch

from oracle import rotations_oracle as R
from sings_amd import camera

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rot_cam_golden.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_product_rotations_refuse_host_tensors():
    """sings_amd.rotations is HIP only: no CPU / eager path behind the product API."""
    import pytest
    from sings_amd import rotations as P
    for fn, w in ((P.quaternion_to_matrix, 4), (P.rotation_6d_to_matrix, 6), (P.axis_angle_to_quaternion, 3),
                  (P.quaternion_to_axis_angle, 4)):
        with pytest.raises(RuntimeError):
            fn(torch.zeros(2, w))
    with pytest.raises(RuntimeError):
        P.matrix_to_quaternion(torch.eye(3)[None])
    with pytest.raises(RuntimeError):
        P.quaternion_multiply(torch.zeros(2, 4), torch.zeros(2, 4))


def test_rotation_conversions_match_reference():
    eq = lambda a, k: np.testing.assert_allclose(a.numpy(), G[k], rtol=1e-6, atol=1e-7)
    eq(R.axis_angle_to_quaternion(T("aa")), "aa_to_q")
    eq(R.axis_angle_to_matrix(T("aa")), "aa_to_m")
    eq(R.axis_angle_to_rotation_6d(T("aa")), "aa_to_d6")
    eq(R.rotation_6d_to_axis_angle(T("d6")), "d6_to_aa")
    eq(R.quaternion_to_axis_angle(T("q")), "q_to_aa")
    eq(R.quaternion_to_matrix(T("q")), "q_to_m")
    eq(R.matrix_to_axis_angle(R.rotation_6d_to_matrix(T("d6"))), "m_to_aa")
    eq(R.matrix_to_rotation_6d(R.rotation_6d_to_matrix(T("d6"))), "m_to_d6")
    eq(R.standardize_quaternion(T("q")), "q_std")
    G1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "lbs_golden.npz"))
    t1 = lambda k: torch.from_numpy(G1[k])
    np.testing.assert_array_equal(R.matrix_to_quaternion(t1("g1_mats")).numpy(), G1["g1_m2q"])
    np.testing.assert_array_equal(R.quaternion_multiply(t1("g1_qa"), t1("g1_qb")).numpy(), G1["g1_qmul"])
    np.testing.assert_allclose(R.rotation_6d_to_matrix(t1("g1_d6")).numpy(), G1["g1_d6_to_mat"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(R.quaternion_to_matrix(t1("g1_qa")).numpy(), G1["g1_q2m"], rtol=1e-6, atol=1e-7)
    # round trip and differentiability
    aa = T("aa")[4:].clone().requires_grad_(True)
    back = R.rotation_6d_to_axis_angle(R.axis_angle_to_rotation_6d(aa))
    m0, m1 = R.axis_angle_to_matrix(aa), R.axis_angle_to_matrix(back)
    assert (m0 - m1).abs().max().item() < 5e-6
    back.sum().backward()
    assert torch.isfinite(aa.grad).all()


def test_orbit_and_static_cameras_match_reference():
    cams = camera.get_rotating_camera(img_size=(896, 512), fov=0.35, dist=4.5, device="cpu", nframes=7)
    for k in ("world_view_transform", "full_proj_transform", "camera_center", "cam_int"):
        np.testing.assert_allclose(np.stack([c[k].numpy() for c in cams]), G["orbit_" + k], rtol=2e-6, atol=2e-6)
    assert cams[0]["image_height"] == 896 and cams[0]["image_width"] == 512
    st = camera.get_static_camera(img_size=256, fov=0.4, device="cpu")
    for k in ("world_view_transform", "full_proj_transform", "camera_center", "cam_int"):
        np.testing.assert_allclose(st[k].numpy(), G["static_" + k], rtol=2e-6, atol=2e-6)
