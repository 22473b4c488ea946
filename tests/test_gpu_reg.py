"""GPU parity of the regulariser kernels (through the C ABI) with golden vectors produced by the reference's classes and
with the CPU oracle at larger sizes.  Tolerances: fp32 sums in a different order -> loss 2e-6 relative, gradients 1e-5
relative + 5e-6 of the gradient scale; k-NN mean edge lengths 2e-6 relative (the SET of the K nearest distances is
exact; sqrt / summation order may differ by an ulp)."""
import os

import numpy as np
import pytest
import torch

from oracle import reg_oracle as ro

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reg_golden.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(a, b, rtol=1e-5, atol_scale=5e-6):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b)
    assert (err <= rtol * np.abs(b) + atol_scale * scale).all(), (err.max(), scale)


def test_region_laplacian_golden():
    from sings_amd.regularizers import RegionLaplacianLoss_v2
    dev = _dev()
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    for tag in ("pos", "col"):
        mod = RegionLaplacianLoss_v2(verts=t("mesh_verts"), edges=t("mesh_edges"), vertex_labels=t("mesh_labels"),
                                     region_weights=G[f"lap_{tag}_w"])
        x = t(f"lap_{tag}_x").requires_grad_(True)
        loss = mod(x)
        (3.0 * loss).backward()                                # upstream scaling goes through autograd
        assert abs(loss.item() - G[f"lap_{tag}_loss"]) <= 2e-6 * abs(G[f"lap_{tag}_loss"])
        _close(x.grad.cpu().numpy() / 3.0, G[f"lap_{tag}_grad"])
        if tag == "pos":
            x2 = t("lap_pos_x").requires_grad_(True)
            lh = mod.forward_hands(x2); lh.backward()
            assert abs(lh.item() - G["lap_hands_loss"]) <= 2e-6 * abs(G["lap_hands_loss"])
            _close(x2.grad.cpu().numpy(), G["lap_hands_grad"])


def test_l2norm_edge_loss_mesh_edge_golden():
    from sings_amd.regularizers import L2Norm, GaussiansEdgeLoss, mesh_edge_loss, knn_mean_edge
    dev = _dev()
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    lam = G["l2_lambdas"]
    mod = L2Norm(lambda_xyz_offsets=float(lam[0]), lambda_scales_diff=float(lam[1]), lambda_max_scale=float(lam[2]),
                 max_scale_threshold=float(lam[3]), lambda_min_opacity=float(lam[4]), min_opacity_threshold=float(lam[5]))
    for tag, keys in (("full", ("xyz_offsets", "scales", "opacity")), ("noop", ("xyz_offsets", "scales"))):
        ins = {'xyz_offsets': t("gs_offsets").requires_grad_(True), 'scales': t("gs_scales").requires_grad_(True),
               'opacity': t("gs_opacity").requires_grad_(True)}
        l = mod({k: ins[k] for k in keys}); l.backward()
        assert abs(l.item() - G[f"l2_{tag}_loss"]) <= 2e-6 * abs(G[f"l2_{tag}_loss"])
        for k in keys:
            _close(ins[k].grad.cpu().numpy(), G[f"l2_{tag}_grad_{k}"])
    sc = t("gs_scales").requires_grad_(True)
    loss = GaussiansEdgeLoss()({'xyz_canon': t("gs_xyz"), 'scales': sc}); loss.backward()
    assert abs(loss.item() - G["edge_loss"]) <= 2e-6 * abs(G["edge_loss"])
    _close(sc.grad.cpu().numpy(), G["edge_grad_scales"])
    # the neighbour search itself: exact K nearest (duplicates and a dense cluster are in the golden point set)
    _, ref_len = ro.gaussians_edge_loss({'xyz_canon': torch.from_numpy(G["gs_xyz"]), 'scales': torch.from_numpy(G["gs_scales"])})
    got = knn_mean_edge(t("gs_xyz")).cpu().numpy()
    np.testing.assert_allclose(got, ref_len.numpy(), rtol=2e-6, atol=1e-9)
    mv = t("mesh_verts").requires_grad_(True)
    ml = mesh_edge_loss(mv, G["mesh_edges"]); ml.backward()
    assert abs(ml.item() - G["mesh_edge_loss"]) <= 2e-6 * abs(G["mesh_edge_loss"])
    _close(mv.grad.cpu().numpy(), G["mesh_edge_grad"])


@pytest.mark.parametrize("N,seed,K", [(30000, 1, 9), (2000, 2, 5), (12000, 3, 17)])
def test_knn_exact_vs_brute_force(N, seed, K):
    """Avatar-like density contrasts (a body-sized sheet, tiny dense clusters, a few far outliers -> multi-ring searches)."""
    from sings_amd.regularizers import knn_mean_edge
    dev = _dev()
    rs = np.random.RandomState(seed)
    xyz = np.stack([rs.uniform(0, 0.6, N), rs.uniform(0, 1.8, N), 0.15 * rs.uniform(-1, 1, N) ** 3], 1).astype(np.float32)
    m = N // 5
    xyz[:m] = (0.004 * rs.normal(size=(m, 3)) + np.array([0.1, 0.9, 0.0])).astype(np.float32)
    xyz[m:m + 5] += np.array([3.0, -2.0, 1.0], np.float32)                         # outliers
    d = torch.cdist(torch.from_numpy(xyz).double(), torch.from_numpy(xyz).double())
    ref = torch.topk(d, K, dim=1, largest=False).values[:, 1:].mean(1).numpy()
    got = knn_mean_edge(torch.from_numpy(xyz).to(dev), K=K).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-6, atol=1e-9)


def test_full_size_properties_150k():
    """cfg4 size (150 k points): deterministic, invariant under a permutation of the points, and scaling the cloud by 2
    doubles every mean edge length exactly (power-of-two scaling is exact in fp32)."""
    from sings_amd.regularizers import knn_mean_edge
    from sings_amd.scene import avatar_scene
    dev = _dev()
    s = avatar_scene(N=150000, J=52)
    x = torch.from_numpy(s["xyz_canon"]).to(dev)
    a = knn_mean_edge(x); b = knn_mean_edge(x)
    assert torch.equal(a, b)
    perm = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(0)).to(dev)
    c = knn_mean_edge(x[perm].contiguous())
    assert torch.allclose(c, a[perm], rtol=1e-6, atol=0)
    d = knn_mean_edge((2.0 * x).contiguous())
    assert torch.allclose(d, 2.0 * a, rtol=1e-6, atol=0)
    assert float(a.min()) >= 0 and torch.isfinite(a).all()
