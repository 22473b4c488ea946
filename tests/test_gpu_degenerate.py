"""Degenerate inputs through the reference-facing operator (GaussianRasterizer; call site gs_renderer_single.py:87-95) against the
CPU oracle: non-finite means / scales / rotations, zero scales, a singular 2-D covariance (App. A.1 step 5: `det == 0 -> return`),
opacities of exactly 0 and 1, every Gaussian on one pixel (equal depths: the stable tie order), 1 x 1 and 17 x 15 images, one SH
row (`M = 1`) at degree 0, a single Gaussian.  The bar is the usual one -- radii bit for bit, RGB <= 1e-5 off borderline pixels,
every gradient at the pytest tolerance -- plus: the call returns (no hang), everything stays finite, and a poisoned Gaussian
changes NOTHING for the others (the image and their gradients are bit-identical to a run without it).

Non-finite inputs have no defined upstream behaviour (its `(int)my_radius` of a NaN is hardware-dependent: 0 on its GPUs ->
radii = 0 and no keys, i.e. nothing rendered); oracle and kernels both CULL such a Gaussian (radius not a positive number below
2^30: oracle/raster_core.inc.c, csrc/sg_project.h) and give it a zero gradient.
"""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from sings_amd.scene import synthetic_scene

pytestmark = pytest.mark.gpu
BORDER = 2e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(name, a, b, rtol=2e-4, atol=2e-6):
    a = np.asarray(a, np.float64).reshape(np.shape(b)); b = np.asarray(b, np.float64)
    assert np.isfinite(a).all(), f"{name}: non-finite values on the HIP side"
    scale = np.abs(b).max() + 1e-30
    bad = np.abs(a - b) > rtol * np.abs(b) + atol * scale
    assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} off; worst {np.abs(a - b).max():.3e} (scale {scale:.3e})"


def _run(s, cov3D=None, colors=None, check=True):
    """HIP forward + backward through the autograd surface, and the oracle on the same arrays.  -> dict of host arrays."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    req = lambda a: None if a is None else t(a).requires_grad_(True)
    rs = GaussianRasterizationSettings(image_height=s["H"], image_width=s["W"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]),
                                       sh_degree=s["sh_degree"], campos=t(s["campos"]), prefiltered=False, debug=False)
    m, op = req(s["means3D"]), req(s["opacities"])
    sh = None if colors is not None else req(s["shs"])
    col = req(colors)
    sc, rt = (None, None) if cov3D is not None else (req(s["scales"]), req(s["rotations"]))
    cv = req(cov3D)
    m2 = torch.zeros_like(m, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=m, means2D=m2, opacities=op, shs=sh, colors_precomp=col, scales=sc, rotations=rt,
                                          cov3D_precomp=cv)
    torch.cuda.synchronize()
    o = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], s["W"], s["H"], s["tanfovx"],
                   s["tanfovy"], s["bg"], scales=None if cov3D is not None else s["scales"],
                   rotations=None if cov3D is not None else s["rotations"], shs=None if colors is not None else s["shs"],
                   sh_degree=s["sh_degree"], colors_precomp=colors, cov3D_precomp=cov3D)
    border = o["margin"] < BORDER
    dL = s["dL_dimage"][:, :s["H"], :s["W"]].copy(); dL[:, border] = 0
    g = ro.backward(o, dL)
    color.backward(t(dL))
    torch.cuda.synchronize()
    c = lambda x: None if x is None or x.grad is None else x.grad.cpu().numpy()
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), o=o, g=g, border=border,
               grads=dict(means3D=c(m), means2D=c(m2), opacity=c(op), sh=c(sh), colors=c(col), scales=c(sc), rotations=c(rt), cov3D=c(cv)))
    if check:
        assert np.isfinite(out["color"]).all()
        np.testing.assert_array_equal(out["radii"], o["radii"])
        diff = np.abs(out["color"] - o["color"]).max(0)
        if (~border).any():
            assert diff[~border].max() <= 1e-5, diff[~border].max()
        if border.any():
            assert (diff[border] - (1e-5 + 1.001 * o["flip"][border])).max() <= 0
        for name, key in (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmean2D"), ("opacity", "dL_dopacity"), ("sh", "dL_dsh"),
                          ("colors", "dL_dcolor"), ("scales", "dL_dscales"), ("rotations", "dL_drots"), ("cov3D", "dL_dcov3D")):
            if out["grads"][name] is not None and g.get(key) is not None:
                _close(name, out["grads"][name], g[key])
    return out


def test_nan_and_inf_means_scales_rotations_are_culled_and_harm_nobody():
    s = synthetic_scene(3000, 160, 96, 2, 31)
    clean = _run(s)
    bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.items()}
    vis = np.nonzero(clean["radii"] > 0)[0]
    pick = vis[:: max(1, vis.size // 60)][:60]                       # 60 Gaussians that are ON the image when healthy
    kinds = {}
    for j, i in enumerate(pick):
        k = j % 6
        kinds[i] = k
        if k == 0: bad["means3D"][i, j % 3] = np.nan
        elif k == 1: bad["means3D"][i, j % 3] = np.inf
        elif k == 2: bad["means3D"][i, 2] = -np.inf
        elif k == 3: bad["scales"][i, j % 3] = np.nan
        elif k == 4: bad["scales"][i, j % 3] = np.inf
        else: bad["rotations"][i, j % 4] = np.nan
    # the oracle on the poisoned arrays (NaN arithmetic on the host is well defined; the explicit radius guard decides the rest)
    with np.errstate(all="ignore"):
        out = _run(bad, check=False)
    o, g = out["o"], out["g"]
    assert (o["radii"][pick] == 0).all() and (out["radii"][pick] == 0).all()
    np.testing.assert_array_equal(out["radii"], o["radii"])
    assert np.isfinite(out["color"]).all()
    for name in ("means3D", "means2D", "opacity", "sh", "scales", "rotations"):
        a = out["grads"][name]
        assert np.isfinite(a).all(), name
        assert np.abs(a[pick]).max() == 0, f"{name}: a culled (non-finite) Gaussian received a gradient"
    # ... and the others see exactly the scene without those 60: delete them and compare BIT FOR BIT
    keep = np.setdiff1d(np.arange(3000), pick)
    sub = dict(s)
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        sub[k] = np.ascontiguousarray(s[k][keep])
    ref = _run(sub)                                                  # (this one also checked against the oracle)
    assert np.array_equal(out["color"], ref["color"])
    for name in ("means3D", "means2D", "opacity", "sh", "scales", "rotations"):
        assert np.array_equal(out["grads"][name][keep], ref["grads"][name]), name
    # the oracle agrees on the healthy ones as well
    border = out["border"]
    assert np.abs(out["color"] - o["color"]).max(0)[~border].max() <= 1e-5
    for name, key in (("means3D", "dL_dmeans3D"), ("opacity", "dL_dopacity"), ("sh", "dL_dsh"), ("scales", "dL_dscales"),
                      ("rotations", "dL_drots"), ("means2D", "dL_dmean2D")):
        _close(name, out["grads"][name][keep], np.nan_to_num(g[key][keep]))


def test_zero_scales_render_the_dilation_kernel():
    """scales = 0: the 3-D covariance vanishes, the 2-D one is the +0.3 dilation alone (radius ceil(3 sqrt(0.3 + sqrt(0.1))) = 3)."""
    s = synthetic_scene(2500, 128, 96, 1, 32)
    s["scales"] = np.zeros_like(s["scales"])
    out = _run(s)
    assert set(np.unique(out["radii"]).tolist()) <= {0, 3} and (out["radii"] == 3).sum() > 1000
    assert np.abs(out["grads"]["scales"]).max() == 0 or np.isfinite(out["grads"]["scales"]).all()


def test_singular_2d_covariance_returns_early():
    """App. A.1 step 5, `det == 0 -> return`, reached exactly: focal 4, an on-axis Gaussian at z = 4 (J = identity block) with
    cov3D_precomp = (-0.3, 0, 0, 1, 0, 1): a = -0.3f + 0.3f = 0, b = 0, c = 1.3 -> det = 0.  Beside it: healthy Gaussians."""
    W = H = 32
    s = synthetic_scene(400, W, H, 0, 33)
    tan = W / (2 * 4.0)
    s.update(tanfovx=tan, tanfovy=tan)
    from sings_amd.camera import get_projection_matrix
    fov = 2 * np.arctan(tan)
    s["projmatrix"] = (s["viewmatrix"] @ get_projection_matrix(0.01, 100.0, fov, fov).T).astype(np.float32)
    o0 = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], W, H, tan, tan, s["bg"],
                    scales=s["scales"], rotations=s["rotations"], colors_precomp=np.full((400, 3), 0.5, np.float32))
    cov = o0["cov3D"].copy()
    # (Gaussians the healthy scene culled have an all-zero stored covariance: fine, they stay culled or render the dilation)
    s["means3D"][:4] = np.array([0, 0, 4.0], np.float32)
    cov[:4] = np.array([-0.3, 0, 0, 1, 0, 1], np.float32)
    colors = np.random.RandomState(3).uniform(0, 1, (400, 3)).astype(np.float32)
    out = _run(s, cov3D=cov, colors=colors)
    assert (out["radii"][:4] == 0).all() and (out["radii"][4:] > 0).sum() > 100
    assert np.abs(out["grads"]["cov3D"][:4]).max() == 0 and np.abs(out["grads"]["means3D"][:4]).max() == 0


def test_opacity_exactly_zero_and_exactly_one():
    s = synthetic_scene(3000, 160, 96, 3, 34)
    s["opacities"][::2] = 0.0
    s["opacities"][1::2] = 1.0
    out = _run(s)
    assert np.abs(out["grads"]["opacity"][::2]).max() == 0            # alpha = 0 < 1/255: skipped everywhere, no gradient
    assert np.abs(out["grads"]["opacity"][1::2]).max() > 0            # the 0.99 clamp is gradient-transparent (App. A.4)


@pytest.mark.parametrize("equal_depths", [False, True])
def test_every_gaussian_on_one_pixel(equal_depths):
    """3000 Gaussians with the same screen position: one tile-list per touched tile of 3000 entries (> 1024: bucket sort); with
    `equal_depths` every key of a list is EQUAL -- the order is then upstream's stable one, by Gaussian index."""
    s = synthetic_scene(3000, 96, 64, 1, 35)
    rs = np.random.RandomState(35)
    z = np.full(3000, 5.0, np.float32) if equal_depths else rs.uniform(3, 8, 3000).astype(np.float32)
    s["means3D"] = np.stack([0.01 * z, -0.02 * z, z], 1).astype(np.float32)        # x / z, y / z constant: one pixel
    s["opacities"] = rs.uniform(0.002, 0.05, (3000, 1)).astype(np.float32)         # faint: the pixel does not saturate early
    out = _run(s)
    tl = out["o"]["ranges"][:, 1].astype(int) - out["o"]["ranges"][:, 0]
    assert tl.max() == 3000
    if equal_depths:
        r0, r1 = out["o"]["ranges"][tl.argmax()]
        assert np.array_equal(out["o"]["point_list"][r0:r1], np.arange(3000))


@pytest.mark.parametrize("W,H", [(1, 1), (17, 15), (16, 16), (33, 1)])
def test_tiny_and_ragged_images(W, H):
    s = synthetic_scene(500, W, H, 2, 36)
    # S() scales splats with the focal length (1.2 W): at W = 1 nothing would touch the pixel -- widen them
    if W < 16:
        s["scales"] = (s["scales"] * 40).astype(np.float32)
    s["dL_dimage"] = np.random.RandomState(1).normal(0, 1, (3, H, W)).astype(np.float32)
    out = _run(s)
    assert out["color"].shape == (3, H, W) and (out["radii"] > 0).any()


def test_one_sh_row_at_degree_zero_and_a_single_gaussian():
    s = synthetic_scene(2000, 96, 64, 0, 37, M=1)
    assert s["shs"].shape == (2000, 1, 3)
    out = _run(s)
    assert out["grads"]["sh"].shape == (2000, 1, 3)
    one = synthetic_scene(1, 64, 48, 3, 38)
    one["means3D"][:] = np.array([0.1, -0.1, 4.0], np.float32); one["scales"][:] = 0.05; one["opacities"][:] = 0.7
    out = _run(one)
    assert out["radii"][0] > 0 and np.abs(out["color"] - one["bg"][:, None, None]).max() > 0.01
    one["means3D"][:] = np.array([0.0, 0.0, -1.0], np.float32)       # ... and the same single Gaussian behind the camera
    out = _run(one)
    assert out["radii"][0] == 0 and np.abs(out["color"] - one["bg"][:, None, None]).max() == 0
