"""CPU: regulariser oracle (oracle/reg_oracle.py) against golden vectors produced by the reference's own classes
(tests/golden/gen_reg_golden.py), and the host-side graph construction of sings_amd/regularizers.py."""
import os

import numpy as np
import torch

from oracle import reg_oracle as ro

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reg_golden.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_region_laplacian_oracle_matches_reference():
    tv, te, tl = T("mesh_verts"), T("mesh_edges"), T("mesh_labels")
    for tag in ("pos", "col"):
        x = T(f"lap_{tag}_x").requires_grad_(True)
        loss = ro.region_laplacian_loss(x, tv, te, tl, G[f"lap_{tag}_w"].astype(np.float64))
        loss.backward()
        assert abs(loss.item() - G[f"lap_{tag}_loss"]) <= 1e-6 * abs(G[f"lap_{tag}_loss"])
        np.testing.assert_allclose(x.grad.numpy(), G[f"lap_{tag}_grad"], rtol=1e-5, atol=1e-9)
    x = T("lap_pos_x").requires_grad_(True)
    lh = ro.region_laplacian_hands(x, tv, te, tl)
    lh.backward()
    assert abs(lh.item() - G["lap_hands_loss"]) <= 1e-6 * abs(G["lap_hands_loss"])
    np.testing.assert_allclose(x.grad.numpy(), G["lap_hands_grad"], rtol=1e-5, atol=1e-9)


def test_l2norm_and_edge_loss_oracle_match_reference():
    lam = G["l2_lambdas"]
    kw = dict(lambda_xyz_offsets=float(lam[0]), lambda_scales_diff=float(lam[1]), lambda_max_scale=float(lam[2]),
              max_scale_threshold=float(lam[3]), lambda_min_opacity=float(lam[4]), min_opacity_threshold=float(lam[5]))
    for tag, keys in (("full", ("xyz_offsets", "scales", "opacity")), ("noop", ("xyz_offsets", "scales"))):
        ins = {'xyz_offsets': T("gs_offsets").requires_grad_(True), 'scales': T("gs_scales").requires_grad_(True),
               'opacity': T("gs_opacity").requires_grad_(True)}
        l = ro.l2norm({k: ins[k] for k in keys}, **kw)
        l.backward()
        assert abs(l.item() - G[f"l2_{tag}_loss"]) <= 1e-6 * abs(G[f"l2_{tag}_loss"])
        for k in keys:
            np.testing.assert_allclose(ins[k].grad.numpy(), G[f"l2_{tag}_grad_{k}"], rtol=1e-5, atol=1e-10)
    sc = T("gs_scales").requires_grad_(True)
    loss, _ = ro.gaussians_edge_loss({'xyz_canon': T("gs_xyz"), 'scales': sc})
    loss.backward()
    assert abs(loss.item() - G["edge_loss"]) <= 1e-6 * abs(G["edge_loss"])
    np.testing.assert_allclose(sc.grad.numpy(), G["edge_grad_scales"], rtol=1e-5, atol=1e-12)


def test_host_csr_construction():
    from sings_amd.regularizers import _csr
    row_ptr, col = _csr(5, np.array([[0, 1], [1, 2], [0, 2], [3, 4]]))
    assert row_ptr.tolist() == [0, 2, 4, 6, 7, 8]
    assert col.tolist() == [1, 2, 0, 2, 0, 1, 4, 3]


def test_cot_laplacian_against_hand_computed_cotangents():
    """oracle.reg_oracle.cot_laplacian (restating pytorch3d's: absent from this image) on triangles whose angles are known:
    weight of an edge = sum over its (<= 2) faces of cot(opposite angle); no diagonal; inv_areas = 1 / sum of adjacent areas."""
    # one right isosceles triangle: angles 90 (at v0), 45, 45
    v = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0]])
    L, inv = ro.cot_laplacian(v, torch.tensor([[0, 1, 2]]))
    assert torch.allclose(L, torch.tensor([[0., 1, 1], [1, 0, 0], [1, 0, 0]]), atol=1e-6)      # edge (1,2) faces the right angle: cot = 0
    assert torch.allclose(inv, torch.full((3, 1), 2.0), atol=1e-6)                              # area 1/2
    # unit square split along (0,2): the diagonal faces two right angles (0 + 0), every side one 45-degree angle (cot 1)
    v = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
    L, inv = ro.cot_laplacian(v, torch.tensor([[0, 1, 2], [0, 2, 3]]))
    exp = torch.zeros(4, 4)
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 0)):
        exp[a, b] = exp[b, a] = 1.0
    assert torch.allclose(L, exp, atol=1e-6) and torch.allclose(L, L.t())
    assert torch.allclose(inv.view(-1), torch.tensor([1.0, 2.0, 1.0, 2.0]), atol=1e-6)
    # equilateral triangle: cot 60 = 1 / sqrt 3 on every edge
    v = torch.tensor([[0., 0, 0], [1, 0, 0], [0.5, 3 ** 0.5 / 2, 0]])
    L, _ = ro.cot_laplacian(v, torch.tensor([[0, 1, 2]]))
    assert torch.allclose(L[L > 0], torch.full((6,), 3 ** -0.5), atol=1e-6)
    # on a closed mesh: symmetric, zero diagonal, and sum_j L_ij (x_j - x_i) of a LINEAR function is the area-weighted zero of the
    # discrete Laplace-Beltrami operator only on flat patches -- here just the structural facts
    from mesh_case import bumpy_sphere
    vv, ff, _ = bumpy_sphere()
    L, _ = ro.cot_laplacian(torch.from_numpy(vv), torch.from_numpy(ff))
    assert torch.allclose(L, L.t(), atol=1e-5) and float(L.diagonal().abs().max()) == 0.0
    assert int((L != 0).sum()) == 2 * (len(ff) * 3 // 2)          # every edge of a closed manifold mesh, both directions
