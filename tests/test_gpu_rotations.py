"""sings_amd.rotations (HIP kernels, csrc/sg_rot.h) against the golden vectors the REFERENCE's rotations.py produced
(tests/golden/rot_cam_golden.npz, golden G1 of lbs_golden.npz) and, for the gradients, against torch.autograd of the
oracle restatement (oracle/rotations_oracle.py, pinned by the same vectors on the CPU) in fp64."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
G = np.load(os.path.join(HERE, "golden", "rot_cam_golden.npz"))
G1 = np.load(os.path.join(HERE, "golden", "lbs_golden.npz"))


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_values_against_the_references_own_outputs():
    from sings_amd import rotations as R
    dev = _dev()
    T = lambda k, g=G: torch.from_numpy(g[k]).to(dev)
    eq = lambda a, ref, tol=2e-6: np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=tol, atol=tol)
    eq(R.axis_angle_to_quaternion(T("aa")), G["aa_to_q"])
    eq(R.axis_angle_to_matrix(T("aa")), G["aa_to_m"])
    eq(R.axis_angle_to_rotation_6d(T("aa")), G["aa_to_d6"])
    eq(R.rotation_6d_to_axis_angle(T("d6")), G["d6_to_aa"], 5e-6)
    eq(R.quaternion_to_axis_angle(T("q")), G["q_to_aa"], 5e-6)
    eq(R.quaternion_to_matrix(T("q")), G["q_to_m"])
    eq(R.matrix_to_axis_angle(R.rotation_6d_to_matrix(T("d6"))), G["m_to_aa"], 5e-6)
    eq(R.matrix_to_rotation_6d(R.rotation_6d_to_matrix(T("d6"))), G["m_to_d6"])
    eq(R.standardize_quaternion(T("q")), G["q_std"], 0)
    # golden G1 (256 seeded inputs incl. non-orthonormal matrices / non-unit quaternions)
    eq(R.rotation_6d_to_matrix(T("g1_d6", G1)), G1["g1_d6_to_mat"])
    eq(R.quaternion_to_matrix(T("g1_qa", G1)), G1["g1_q2m"])
    np.testing.assert_allclose(R.quaternion_multiply(T("g1_qa", G1), T("g1_qb", G1)).cpu().numpy(), G1["g1_qmul"], rtol=1e-6, atol=1e-7)
    # leading batch dimensions, broadcasting of quaternion_multiply (ext_tfs: one rotation times N, sings_hybrid.py:421-428)
    assert R.quaternion_to_matrix(T("q").view(10, 30, 4)).shape == (10, 30, 3, 3)
    one = T("g1_qa", G1)[:1]
    many = T("g1_qb", G1)
    np.testing.assert_array_equal(R.quaternion_multiply(one, many).cpu().numpy(),
                                  R.quaternion_multiply(one.expand_as(many).contiguous(), many).cpu().numpy())


@pytest.mark.parametrize("name,width", [("quaternion_to_matrix", 4), ("rotation_6d_to_matrix", 6),
                                        ("axis_angle_to_quaternion", 3), ("quaternion_to_axis_angle", 4),
                                        ("axis_angle_to_matrix", 3), ("rotation_6d_to_axis_angle", 6),
                                        ("matrix_to_axis_angle", 9)])
def test_gradients_against_autograd_of_the_oracle(name, width):
    from oracle import rotations_oracle as RO
    from sings_amd import rotations as R
    dev = _dev()
    rs = np.random.RandomState(len(name) * 31 + width)
    n = 4096
    if name == "matrix_to_axis_angle":
        x = RO.axis_angle_to_matrix(torch.from_numpy((1.5 * rs.randn(n, 3)).astype(np.float32))).numpy()
    else:
        x = rs.randn(n, width).astype(np.float32)
        if name == "quaternion_to_axis_angle":
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            x[:, 0] = np.abs(x[:, 0])                                            # away from the atan2 branch cut at angle 2 pi
    ref_in = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    out_ref = getattr(RO, name)(ref_in)
    g = rs.randn(*out_ref.shape)
    (out_ref * torch.from_numpy(g)).sum().backward()
    a = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = getattr(R, name)(a)
    (out * torch.from_numpy(g.astype(np.float32)).to(dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=3e-5, atol=3e-5)
    gr, gg = ref_in.grad.numpy(), a.grad.cpu().numpy().astype(np.float64)
    scale = np.abs(gr).max()
    bad = np.abs(gg - gr) > 2e-4 * np.abs(gr) + 2e-5 * scale
    assert bad.mean() < 2e-3, f"{name}: {bad.sum()} of {bad.size} gradient entries off (max {np.abs(gg - gr).max()})"


def test_quaternion_multiply_gradient_and_small_angles():
    from oracle import rotations_oracle as RO
    from sings_amd import rotations as R
    dev = _dev()
    rs = np.random.RandomState(9)
    a_n, b_n, g_n = rs.randn(3000, 4), rs.randn(3000, 4), rs.randn(3000, 4)
    ar = torch.from_numpy(a_n).requires_grad_(True); br = torch.from_numpy(b_n).requires_grad_(True)
    (RO.quaternion_multiply(ar, br) * torch.from_numpy(g_n)).sum().backward()
    a = torch.from_numpy(a_n.astype(np.float32)).to(dev).requires_grad_(True)
    b = torch.from_numpy(b_n.astype(np.float32)).to(dev).requires_grad_(True)
    (R.quaternion_multiply(a, b) * torch.from_numpy(g_n.astype(np.float32)).to(dev)).sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(b.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-5)
    # the reference's small-angle series (|angle| < 1e-6) and the zero vector: finite values and gradients
    tiny = torch.tensor([[0.0, 0.0, 0.0], [3e-7, -2e-7, 1e-7], [1e-3, 0.0, 0.0]], device=dev, requires_grad=True)
    q = R.axis_angle_to_quaternion(tiny)
    np.testing.assert_allclose(q.detach().cpu().numpy(), RO.axis_angle_to_quaternion(tiny.detach().cpu()).numpy(), rtol=1e-6, atol=1e-9)
    q.sum().backward()
    ref = tiny.detach().cpu().double().requires_grad_(True)
    RO.axis_angle_to_quaternion(ref).sum().backward()
    np.testing.assert_allclose(tiny.grad.cpu().numpy(), ref.grad.numpy(), rtol=1e-4, atol=1e-6)
    back = R.quaternion_to_axis_angle(q.detach())
    np.testing.assert_allclose(back.cpu().numpy(), tiny.detach().cpu().numpy(), rtol=1e-4, atol=1e-9)
