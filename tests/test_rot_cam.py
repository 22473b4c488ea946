"""CPU: host-side rotation conversions (sings_amd/rotations.py) and orbit / static cameras (sings_amd/camera.py) against
golden vectors produced by the reference's own functions (tests/golden/gen_rot_cam_golden.py)."""
import os

import numpy as np
import torch

from sings_amd import camera, rotations as R

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rot_cam_golden.npz"))
T = lambda k: torch.from_numpy(G[k])


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
