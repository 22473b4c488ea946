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
