"""the two independent restatements of the (unpinnable) upstream rasterizer against each other --
oracle/raster_core.inc.c (scalar C, hand-derived explicit backward) vs oracle/raster_torch.py (vectorised torch, gradients
by autograd).  Integer outcomes exactly, images and gradients in fp64 to round-off."""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from oracle import raster_torch as rt
from sings_amd.scene import synthetic_scene

CASES = {"a": (400, 96, 64, 3, 101), "b": (1500, 80, 112, 1, 102), "c": (60, 33, 17, 0, 103)}     # tests/golden/gen_raster_golden.py


def _c_oracle(s, dL, dtype, mod=1.0):
    o = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], s["W"], s["H"], s["tanfovx"],
                   s["tanfovy"], s["bg"], scales=s["scales"], rotations=s["rotations"], shs=s["shs"], sh_degree=s["sh_degree"],
                   scale_modifier=mod, dtype=dtype)
    return o, ro.backward(o, dL)


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("tag", sorted(CASES))
@pytest.mark.parametrize("mod", [1.0, 0.6])
def test_c_restatement_and_autograd_twin_agree(tag, mod):
    N, W, H, deg, seed = CASES[tag]
    s = synthetic_scene(N, W, H, deg, seed)
    dL = s["dL_dimage"].astype(np.float64)
    o, g = _c_oracle(s, dL, np.float64, mod)
    t, tg = rt.forward_backward(s, dL, torch.float64, scale_modifier=mod)
    # integer outcomes: identical
    np.testing.assert_array_equal(t["radii"], o["radii"])
    vis = o["radii"] > 0
    np.testing.assert_array_equal(t["rect"][vis], o["rect"][vis])
    np.testing.assert_array_equal(t["tiles_touched"], o["tiles_touched"])
    assert t["R"] == o["R"]
    np.testing.assert_array_equal(t["keys"], o["keys"])
    np.testing.assert_array_equal(t["point_list"], o["point_list"])
    np.testing.assert_array_equal(t["ranges"], o["ranges"])
    np.testing.assert_array_equal(t["n_contrib"], o["n_contrib"])
    # images / per-Gaussian records
    assert _rel(t["color"], o["color"]) < 1e-12
    assert _rel(t["final_T"].numpy(), o["final_T"]) < 1e-12
    assert _rel(t["xy"].numpy()[vis], o["xy"][vis]) < 1e-13 and _rel(t["rgb"].numpy()[vis], o["rgb"][vis]) < 1e-13
    assert _rel(t["conic"].numpy()[vis], o["conic_opacity"][vis, :3]) < 1e-12
    # every gradient: autograd of the twin's forward == the C file's explicit backward
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drots", "dL_dsh"):
        assert _rel(tg[k], g[k].reshape(tg[k].shape)) < 2e-9, (k, _rel(tg[k], g[k].reshape(tg[k].shape)))
    assert _rel(tg["dL_dmean2D"], g["dL_dmean2D"][:, :2]) < 2e-9


def test_twin_in_fp32_tracks_the_fp32_checker():
    """The precision the HIP path is checked in: the fp32 run of the twin against the fp32 C oracle (different operation
    order inside the covariance algebra, so round-off apart; pixels near a hard threshold excluded like everywhere else)."""
    N, W, H, deg, seed = CASES["a"]
    s = synthetic_scene(N, W, H, deg, seed)
    o, _ = _c_oracle(s, s["dL_dimage"], np.float32)
    t = rt.rasterize(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], W, H, s["tanfovx"], s["tanfovy"],
                     s["bg"], s["scales"], s["rotations"], s["shs"], deg, dtype=torch.float32)
    vis = o["radii"] > 0
    assert vis.sum() > 300
    np.testing.assert_array_equal(t["depths"].numpy().view(np.uint32)[vis], o["depths"].view(np.uint32)[vis])   # depth key bits
    strict = o["margin"] >= 2e-5
    same = t["n_contrib"] == o["n_contrib"]
    assert np.abs(t["color"].detach().numpy() - o["color"]).max(0)[strict & same].max() < 1e-5
    assert (strict & same).mean() > 0.95


def test_scale_gradient_is_reported_for_the_modified_scale():
    """Upstream's computeCov3D backward returns dL/d(mod * scale) as dL/dscale -- no factor mod (a known property of the
    upstream kernels, moot at mod = 1).  Both restatements implement that; the derivative of the forward itself is mod x."""
    N, W, H, deg, seed = CASES["c"]
    s = synthetic_scene(N, W, H, deg, seed)
    dL = s["dL_dimage"].astype(np.float64)
    mod, eps = 0.6, 1e-6
    o, g = _c_oracle(s, dL, np.float64, mod)
    i = int(np.argmax(np.abs(g["dL_dscales"]).sum(1)))
    def loss(scales):
        s2 = dict(s); s2["scales"] = scales
        oo = ro.forward(s2["means3D"], s2["opacities"], s2["viewmatrix"], s2["projmatrix"], s2["campos"], W, H, s2["tanfovx"],
                        s2["tanfovy"], s2["bg"], scales=scales, rotations=s2["rotations"], shs=s2["shs"], sh_degree=deg,
                        scale_modifier=mod, dtype=np.float64)
        return float((oo["color"] * dL).sum())
    sc = s["scales"].astype(np.float64)
    up, dn = sc.copy(), sc.copy()
    up[i, 0] += eps; dn[i, 0] -= eps
    fd = (loss(up) - loss(dn)) / (2 * eps)
    assert abs(fd - mod * g["dL_dscales"][i, 0]) <= 1e-5 * abs(fd) + 1e-9
