"""GPU parity of the stand-alone lbs_extra: sings_amd.lbs.lbs_extra -- the reference's signature and return
values (sings/rec/utils/body_model/lbs.py:16-74) -- against golden G3 (the reference's own lbs_extra, tests/golden/
gen_lbs_golden.py) and, for gradients / batches / the pose-corrective path, the torch-CPU oracle in fp64."""
import os

import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lbs_golden.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("J", [24, 52])
def test_lbs_extra_golden(J):
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A, w, v = t(G[f"g3_A_{J}"]), t(G[f"g3_w_{J}"]), t(G[f"g3_v_{J}"])
    verts, A_out, T, v_posed, v_shaped = lbs_extra(A[None], v[None], None, w, torch.zeros(1, J * 3, device=dev), disable_posedirs=True,
                                                   pose2rot=True)
    assert verts.shape == (1, 1000, 3) and T.shape == (1, 1000, 4, 4) and A_out is not None and v_posed.shape == (1, 1000, 3)
    # the reference sums the J (and 4) products in rocBLAS / MKL order, the kernel in joint order on the matrix cores
    np.testing.assert_allclose(T[0].cpu().numpy(), G[f"g3_T_{J}"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(verts[0].cpu().numpy(), G[f"g3_verts_{J}"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("N,J,B", [(3001, 52, 2), (64, 24, 1), (1, 3, 1), (777, 64, 1)])
def test_lbs_extra_gradients_and_batches(N, J, B):
    """verts, T and the gradients w.r.t. A and v (upstream gradients on BOTH outputs) against the fp64 oracle; N not a
    multiple of the 64-point waves, J not a multiple of 4 / 16, J = 64 (the maximum), a batch of frames."""
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    rs = np.random.RandomState(N + J)
    A_n = (np.eye(4)[None, None] + 0.3 * rs.randn(B, J, 4, 4)).astype(np.float32)
    w_n = rs.rand(N, J).astype(np.float32) ** 4; w_n /= w_n.sum(1, keepdims=True)
    v_n = rs.randn(B, N, 3).astype(np.float32)
    gv, gT = rs.randn(B, N, 3).astype(np.float32), rs.randn(B, N, 4, 4).astype(np.float32)
    A = torch.from_numpy(A_n).to(dev).requires_grad_(True); v = torch.from_numpy(v_n).to(dev).requires_grad_(True)
    verts, _, T, _, _ = lbs_extra(A, v, None, torch.from_numpy(w_n).to(dev), None, disable_posedirs=True)
    ((verts * torch.from_numpy(gv).to(dev)).sum() + (T * torch.from_numpy(gT).to(dev)).sum()).backward()
    A64 = torch.from_numpy(A_n).double().requires_grad_(True); v64 = torch.from_numpy(v_n).double().requires_grad_(True)
    vo, To = lo.lbs_extra(A64, v64, torch.from_numpy(w_n).double())
    ((vo * torch.from_numpy(gv).double()).sum() + (To * torch.from_numpy(gT).double()).sum()).backward()

    def close(a, b, what):
        a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().numpy()
        assert (np.abs(a - b) <= 2e-5 * np.abs(b) + 2e-6 * np.abs(b).max()).all(), (what, np.abs(a - b).max())
    close(verts, vo, "verts"); close(T, To, "T"); close(v.grad, v64.grad, "dv"); close(A.grad, A64.grad, "dA")
    # only one of the two outputs used downstream (the other gradient is None / zero)
    A.grad = None; v.grad = None
    verts2, _, T2, _, _ = lbs_extra(A, v, None, torch.from_numpy(w_n).to(dev), None, disable_posedirs=True)
    (T2 * torch.from_numpy(gT).to(dev)).sum().backward()
    A64.grad = None; v64.grad = None
    vo, To = lo.lbs_extra(A64, v64, torch.from_numpy(w_n).double())
    (To * torch.from_numpy(gT).double()).sum().backward()
    close(A.grad, A64.grad, "dA (T only)")
    assert float(v.grad.abs().max()) == 0.0


def test_lbs_extra_pose_correctives_and_errors():
    """disable_posedirs=False: v_posed = v_shaped + pose_feature @ posedirs (lbs.py:27-36) before the skinning."""
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    rs = np.random.RandomState(3)
    N, J = 500, 24
    A = torch.from_numpy((np.eye(4)[None, None] + 0.2 * rs.randn(1, J, 4, 4)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rs.rand(N, J).astype(np.float32)).to(dev); w = w / w.sum(1, keepdim=True)
    v = torch.from_numpy(rs.randn(1, N, 3).astype(np.float32)).to(dev)
    posedirs = torch.from_numpy((0.01 * rs.randn((J - 1) * 9, N * 3)).astype(np.float32)).to(dev)
    pose = torch.from_numpy((0.3 * rs.randn(1, J * 3)).astype(np.float32)).to(dev)
    verts, _, T, v_posed, v_shaped = lbs_extra(A, v, posedirs, w, pose, disable_posedirs=False, pose2rot=True)
    R = lo.batch_rodrigues(pose.cpu().double().view(-1, 3)).view(1, J, 3, 3)
    off = ((R[:, 1:] - torch.eye(3, dtype=torch.float64)).reshape(1, -1) @ posedirs.cpu().double()).view(1, N, 3)
    vo, To = lo.lbs_extra(A.cpu().double(), v.cpu().double() + off, w.cpu().double())
    np.testing.assert_allclose(v_posed.cpu().numpy(), (v.cpu().double() + off).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(verts.cpu().numpy(), vo.numpy(), rtol=2e-5, atol=2e-5)
    assert v_shaped is v
    with pytest.raises(RuntimeError):
        lbs_extra(A, v, None, w.clone().requires_grad_(True), None, disable_posedirs=True)
    with pytest.raises(RuntimeError):
        lbs_extra(A.cpu(), v.cpu(), None, w.cpu(), None, disable_posedirs=True)
