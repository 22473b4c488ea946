"""CPU: decode oracle (oracle/decode_oracle.py) against golden vectors produced by the reference's own modules."""
import os

import numpy as np
import torch

from oracle import decode_oracle as do

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_golden.npz"))
T = lambda k: torch.from_numpy(G[k])


def _grids(req=False):
    return [[T(f"tp_plane_{s}_{c}").requires_grad_(req) for c in range(3)] for s in range(len(G["tp_multires"]))]


def test_triplane_oracle_matches_reference():
    grids = _grids(True)
    pts = T("tp_pts").requires_grad_(True)
    feats = do.triplane_features(pts, grids, T("tp_aabb"))
    np.testing.assert_allclose(feats.detach().numpy(), G["tp_feats"], rtol=1e-6, atol=1e-9)
    (feats * T("tp_w")).sum().backward()
    np.testing.assert_allclose(pts.grad.numpy(), G["tp_dpts"], rtol=1e-5, atol=1e-7)
    for s in range(len(grids)):
        for c in range(3):
            np.testing.assert_allclose(grids[s][c].grad.numpy(), G[f"tp_dplane_{s}_{c}"], rtol=1e-5, atol=1e-7)


def test_decoder_oracles_match_reference():
    for tag, iso in (("iso", True), ("aniso", False)):
        sd = {k[len(f"geo_{tag}_p_"):]: T(k) for k in G.files if k.startswith(f"geo_{tag}_p_")}
        o = do.geometry_decoder(T(f"geo_{tag}_x"), sd, isotropic=iso)
        for k in ('xyz_offsets', 'scales', 'scales_aux') + (() if iso else ('rotations',)):
            np.testing.assert_allclose(o[k].numpy(), G[f"geo_{tag}_o_{k}"], rtol=1e-5, atol=1e-7)
    sd = {k[len("app_p_"):]: T(k) for k in G.files if k.startswith("app_p_")}
    o = do.appearance_decoder(T("geo_iso_x"), sd, T("app_offset"))
    np.testing.assert_allclose(o['shs'].numpy(), G["app_o_shs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o['opacity'].numpy(), G["app_o_opacity"], rtol=1e-5, atol=1e-7)
