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


def test_cfg5_regularisers_value_and_gradient_at_500k_points():
    """BASELINE configs[4]: the geometry-preserving regulariser backward at 500 k Gaussians (the cfg5 scene's own points and
    scales).  The whole cloud cannot be checked by brute force, so: (i) exactness of the k-NN on a 20 k SUBSAMPLE of the
    same cloud against fp64 brute force; (ii) at full size, determinism, permutation invariance and exact power-of-two
    scaling of the neighbour distances; (iii) GaussiansEdgeLoss + L2Norm value and gradient at 500 k against the closed
    form of loss_items.py:57-90,  loss = mean((s_0 - d)^2),  dL/ds_0 = 2 (s_0 - d) / N  (first scale component only),
    evaluated in fp64 from the kernel's own (verified) neighbour distances, and against the oracle's autograd on the
    subsample."""
    from oracle import reg_oracle as ro
    from sings_amd.regularizers import GaussiansEdgeLoss, L2Norm, knn_mean_edge
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    s = synthetic_scene(500000, 2048, 2048, 3, 5)
    xyz_n, sc_n, op_n = s["means3D"], s["scales"], s["opacities"]
    x = torch.from_numpy(xyz_n).to(dev)
    # (i) subsample: exact against brute force
    sub = np.random.RandomState(0).choice(500000, 20000, replace=False)
    xs = torch.from_numpy(xyz_n[sub])
    d = torch.cdist(xs.double(), xs.double())
    ref = torch.topk(d, 9, dim=1, largest=False).values[:, 1:].mean(1).numpy()
    got = knn_mean_edge(xs.to(dev)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-6, atol=1e-9)
    # (ii) full size: invariances
    a = knn_mean_edge(x)
    assert torch.equal(a, knn_mean_edge(x)) and torch.isfinite(a).all() and float(a.min()) >= 0
    perm = torch.randperm(500000, generator=torch.Generator().manual_seed(1)).to(dev)
    assert torch.allclose(knn_mean_edge(x[perm].contiguous()), a[perm], rtol=1e-6, atol=0)
    assert torch.allclose(knn_mean_edge((0.5 * x).contiguous()), 0.5 * a, rtol=1e-6, atol=0)
    # (iii) value + gradient at full size
    sc = torch.from_numpy(sc_n).to(dev).requires_grad_(True)
    off = (0.002 * torch.randn(500000, 3, generator=torch.Generator().manual_seed(2))).to(dev).requires_grad_(True)
    op = torch.from_numpy(op_n).to(dev).requires_grad_(True)
    edge = GaussiansEdgeLoss()({'xyz_canon': x, 'scales': sc})
    l2 = L2Norm()({'xyz_offsets': off, 'scales': sc, 'opacity': op})
    (edge + l2).backward()
    g_total = sc.grad.clone()
    sc2 = torch.from_numpy(sc_n).to(dev).requires_grad_(True)
    GaussiansEdgeLoss()({'xyz_canon': x, 'scales': sc2}).backward()
    d64 = a.double().cpu(); s64 = torch.from_numpy(sc_n).double()
    s0 = s64[:, 0]
    loss_ref = ((s0 - d64) ** 2).mean()
    assert abs(edge.item() - loss_ref.item()) <= 3e-6 * abs(loss_ref.item())
    g_ref = np.zeros((500000, 3)); g_ref[:, 0] = (2.0 * (s0 - d64) / 500000.0).numpy()
    _close(sc2.grad.cpu().numpy(), g_ref)
    # L2Norm: pure torch in the reference -> the oracle restatement (pinned by reg_golden.npz) on the CPU, full size
    # (in fp64: a fp32 norm over 1.5e6 numbers carries ~1e-5 of rounding itself)
    ins = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in (("xyz_offsets", off), ("scales", sc), ("opacity", op))}
    l2_ref = ro.l2norm(ins)
    l2_ref.backward()
    assert abs(l2.item() - l2_ref.item()) <= 3e-6 * abs(l2_ref.item())
    _close(off.grad.cpu().numpy(), ins["xyz_offsets"].grad.numpy())
    _close(op.grad.cpu().numpy(), ins["opacity"].grad.numpy())
    _close((g_total - sc2.grad).cpu().numpy(), ins["scales"].grad.numpy())
    # subsample: the oracle's own autograd (k-NN restated from its published definition) for value and gradient
    scs = torch.from_numpy(sc_n[sub]).requires_grad_(True)
    lo, _ = ro.gaussians_edge_loss({'xyz_canon': xs, 'scales': scs}); lo.backward()
    scd = torch.from_numpy(sc_n[sub]).to(dev).requires_grad_(True)
    ld = GaussiansEdgeLoss()({'xyz_canon': xs.to(dev), 'scales': scd}); ld.backward()
    assert abs(ld.item() - lo.item()) <= 3e-6 * abs(lo.item())
    _close(scd.grad.cpu().numpy(), scs.grad.numpy())


def test_edge_loss_without_gradients_returns_the_value_not_an_unwritten_buffer():
    """GaussiansEdgeLoss.prepare() attached the autograd node BEFORE the query kernel had written the loss; with no
    input that requires grad (evaluation under no_grad, detached scales) it returned a clone of the uninitialised buffer.
    Every other test uses requires_grad=True.  The value must equal the golden loss in all three modes, and a second
    prepare() without finish() must raise instead of orphaning the first node."""
    from sings_amd.regularizers import GaussiansEdgeLoss
    dev = _dev()
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    want = float(G["edge_loss"])
    mod = GaussiansEdgeLoss()
    junk = torch.full((1 << 20,), 1e30, device=dev); del junk        # whatever the allocator hands out next is not zero
    with torch.no_grad():
        l0 = mod({'xyz_canon': t("gs_xyz"), 'scales': t("gs_scales").requires_grad_(True)})
    l1 = mod({'xyz_canon': t("gs_xyz"), 'scales': t("gs_scales")})                   # nothing requires grad
    l2 = mod({'xyz_canon': t("gs_xyz"), 'scales': t("gs_scales").requires_grad_(True).detach()})
    for l in (l0, l1, l2):
        assert not l.requires_grad and abs(l.item() - want) <= 2e-6 * abs(want), (l.item(), want)
    out = mod.prepare({'xyz_canon': t("gs_xyz"), 'scales': t("gs_scales")})
    with pytest.raises(RuntimeError, match="twice"):
        mod.prepare({'xyz_canon': t("gs_xyz"), 'scales': t("gs_scales")})
    mod.finish()
    assert abs(out.item() - want) <= 2e-6 * abs(want)


def test_region_laplacian_cotangent_vs_oracle():
    """RegionLaplacianLoss_v2(laplacian_type='cotangent') (loss_items.py:150-165: overlapping regions, cot weights, no diagonal)
    through sg_rows_laplacian against oracle.reg_oracle.region_laplacian_cot_loss + autograd, forward and forward_hands;
    'norm' raises as in the reference (:110-112), 'cotangent' without faces raises as in :130-131."""
    from sings_amd.regularizers import RegionLaplacianLoss_v2
    from mesh_case import bumpy_sphere
    dev = _dev()
    vv, ff, lab = bumpy_sphere(nu=48, nv=40, seed=3)
    rng = np.random.default_rng(5)
    w = rng.uniform(0.5, 2.0, 8)
    for C_ in (3, 1):
        x = rng.standard_normal((len(vv), C_)).astype(np.float32)
        xo = torch.from_numpy(x).requires_grad_(True)
        lo = ro.region_laplacian_cot_loss(xo, torch.from_numpy(vv), torch.from_numpy(ff), torch.from_numpy(lab), w)
        lo.backward()
        mod = RegionLaplacianLoss_v2(verts=torch.from_numpy(vv).to(dev), edges=None, vertex_labels=torch.from_numpy(lab).to(dev),
                                     faces=torch.from_numpy(ff).to(dev), region_weights=w, laplacian_type="cotangent")
        xg = torch.from_numpy(x).to(dev).requires_grad_(True)
        lg = mod(xg)
        (2.0 * lg).backward()
        assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item()), (lg.item(), lo.item())
        _close(xg.grad.cpu().numpy() / 2.0, xo.grad.numpy(), rtol=2e-5, atol_scale=1e-5)
        xo2 = torch.from_numpy(x).requires_grad_(True)
        lh = ro.region_laplacian_cot_loss(xo2, torch.from_numpy(vv), torch.from_numpy(ff), torch.from_numpy(lab), w, only=(6, 7), strength=1000)
        lh.backward()
        xg2 = torch.from_numpy(x).to(dev).requires_grad_(True)
        lhg = mod.forward_hands(xg2); lhg.backward()
        assert abs(lhg.item() - lh.item()) <= 1e-5 * abs(lh.item())
        _close(xg2.grad.cpu().numpy(), xo2.grad.numpy(), rtol=2e-5, atol_scale=1e-5)
    with pytest.raises(NotImplementedError):
        RegionLaplacianLoss_v2(verts=torch.from_numpy(vv).to(dev), edges=None, vertex_labels=torch.from_numpy(lab).to(dev),
                               faces=torch.from_numpy(ff).to(dev), laplacian_type="norm")
    with pytest.raises(ValueError):
        RegionLaplacianLoss_v2(verts=torch.from_numpy(vv).to(dev), edges=None, vertex_labels=torch.from_numpy(lab).to(dev),
                               laplacian_type="cotangent")
