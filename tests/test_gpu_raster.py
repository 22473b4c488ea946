"""GPU parity: HIP rasterizer (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): bit-exact tile/key indices (radii, tile rectangles, depth key
bits, sorted point list, tile ranges, upstream-format keys); per-pixel RGB within 1e-5 of the
oracle; gradients within a relative tolerance (fp32 summation order differs by design).

The oracle evaluates exp() with libm, the kernel with v_exp_f32 (1 ulp).  A pixel whose
alpha / transmittance lands within 2e-5 (relative) of one of the hard thresholds of the algorithm
(alpha < 1/255, T < 1e-4, power > 0) can therefore legitimately flip; the oracle reports that
margin per pixel and such pixels (a few per 10^5 pixels) are compared with a loose bound
and counted.
"""
import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from sings_amd.scene import synthetic_scene

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-5
BORDER = 2e-5          # relative threshold margin below which a pixel is "borderline" (exp noise ~2e-7)


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _settings(s, dev, debug=False, scale_modifier=1.0):
    from sings_amd.rasterizer import GaussianRasterizationSettings
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return GaussianRasterizationSettings(
        image_height=s["H"], image_width=s["W"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
        scale_modifier=scale_modifier, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]),
        sh_degree=s["sh_degree"], campos=t(s["campos"]), prefiltered=False, debug=debug)


def _oracle(s, scale_modifier=1.0, colors=None):
    return ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], s["W"], s["H"],
                      s["tanfovx"], s["tanfovy"], s["bg"], scales=s["scales"], rotations=s["rotations"],
                      shs=None if colors is not None else s["shs"], sh_degree=s["sh_degree"],
                      colors_precomp=colors, scale_modifier=scale_modifier)


def _check_forward_state(s, st, o):
    vis = o["radii"] > 0
    np.testing.assert_array_equal(st["radii"].cpu().numpy(), o["radii"])
    # depth key bits, pixel centres, conics, colours: same fp32 operation order -> bit exact
    for name, a, b in (("depths", st["depths"], o["depths"]), ("xy", st["xy"], o["xy"]),
                       ("conic_opacity", st["conic_opacity"], o["conic_opacity"]), ("rgb", st["rgb"], o["rgb"])):
        a = a.cpu().numpy()[vis]; b = b[vis]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            f"{name}: {np.sum(a.view(np.uint32) != b.view(np.uint32))} of {a.size} words differ, max abs {np.abs(a-b).max()}"
    rmin = st["rect_min"].cpu().numpy(); rwh = st["rect_wh"].cpu().numpy()
    rect = np.stack([rmin & 0xffff, rmin >> 16, (rmin & 0xffff) + (rwh & 0xffff), (rmin >> 16) + (rwh >> 16)], 1)
    np.testing.assert_array_equal(rect[vis], o["rect"][vis])
    clamp = st["clamp_bits"].cpu().numpy()
    exp_bits = o["clamped"][:, 0] | (o["clamped"][:, 1] << 1) | (o["clamped"][:, 2] << 2)
    np.testing.assert_array_equal(clamp[vis], exp_bits[vis])
    assert st["R"] == o["R"]
    np.testing.assert_array_equal(st["ranges"].cpu().numpy().astype(np.uint32), o["ranges"])
    np.testing.assert_array_equal(st["point_list"].cpu().numpy().astype(np.uint32), o["point_list"])
    np.testing.assert_array_equal(st["point_keys"].cpu().numpy().astype(np.uint64), o["keys"])


BORDERLINE = []        # (test id, borderline pixels, pixels, > 1e-5 off, worst borderline |delta|): report by conftest.pytest_terminal_summary


def _check_image(color, final_T, n_contrib, o):
    """RGB <= 1e-5 on every pixel whose hard-threshold decisions (alpha >= 1/255, T >= 1e-4, power <= 0) have a relative
    margin >= 2e-5 in the oracle.  The others ("borderline": fp32 exp noise can flip the decision) may differ by the
    contribution of the splat(s) whose decision is borderline -- the oracle bounds that per pixel (``o["flip"]``:
    alpha T (|c| + cmax) per borderline alpha / power decision, 2 T cmax per borderline termination) and the pixel must stay
    within 1e-5 + that bound; their NUMBER and the WORST borderline error are recorded per test (printed at the end of the
    run by conftest.py).  Observed on the MI355X: ~150 of 2.0 M pixels across the suite, one of them beyond 1e-5."""
    import os
    diff = np.abs(color - o["color"]).max(0)
    border = o["margin"] < BORDER
    nb = int(border.sum())
    worst = float(diff[border].max()) if nb else 0.0
    BORDERLINE.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], nb, int(border.size),
                       int((diff[border] > RGB_TOL).sum()) if nb else 0, worst))
    assert nb <= 8 + 6e-4 * border.size, f"{nb} borderline pixels of {border.size}"
    strict = ~border
    assert diff[strict].max() <= RGB_TOL, f"RGB L_inf {diff[strict].max()} at {np.argwhere(diff == diff[strict].max())[:3]}"
    if nb:
        over = diff[border] - (RGB_TOL + 1.001 * o["flip"][border])
        assert over.max() <= 0, (f"a borderline pixel is off by {diff[border][over.argmax()]:.3e}, more than the contribution of its "
                                 f"borderline splat(s) ({o['flip'][border][over.argmax()]:.3e})")
    assert np.abs(final_T - o["final_T"])[strict].max() <= 1e-5
    assert np.array_equal(n_contrib[strict].astype(np.uint32), o["n_contrib"][strict])
    return nb


def _grad_close(name, a, b, rtol=2e-4, atol=2e-6):
    a = a.astype(np.float64); b = b.astype(np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b)
    bound = rtol * np.abs(b) + atol * scale
    bad = err > bound
    assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} off; worst abs {err.max():.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("N,W,H,deg,seed", [(2000, 128, 128, 3, 1), (3000, 200, 136, 1, 11), (500, 64, 48, 0, 5)])
def test_forward_state_and_image(N, W, H, deg, seed):
    from sings_amd.inspect_ws import forward_with_state
    dev = _dev()
    s = synthetic_scene(N, W, H, deg, seed)
    o = _oracle(s)
    rs = _settings(s, dev, debug=True)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                            rotations=t(s["rotations"]))
    _check_forward_state(s, st, o)
    _check_image(st["color"].cpu().numpy(), st["final_T"].cpu().numpy(), st["n_contrib"].cpu().numpy(), o)


@pytest.mark.parametrize("mod", [1.0, 0.5, 1.7])
def test_forward_backward_autograd_api(mod):
    """GaussianRasterizer (the reference-facing API) forward + backward vs oracle, at scale_modifier = 1 and != 1 (the
    scale gradient is then the one upstream reports: w.r.t. the modified scale, no factor scale_modifier)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(4000, 160, 128, 3, 21)
    o = _oracle(s, scale_modifier=mod)
    g = ro.backward(o, s["dL_dimage"])
    rs = _settings(s, dev, scale_modifier=mod)
    t = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    means3D, opac, shs, scales, rots = t(s["means3D"]), t(s["opacities"]), t(s["shs"]), t(s["scales"]), t(s["rotations"])
    means2D = torch.zeros_like(means3D, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                          scales=scales, rotations=rots)
    assert radii.dtype == torch.int32 and not radii.requires_grad
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    border = o["margin"] < BORDER
    diff = np.abs(color.detach().cpu().numpy() - o["color"]).max(0)
    assert diff[~border].max() <= RGB_TOL
    dL = torch.from_numpy(s["dL_dimage"]).to(dev)
    # pixels whose threshold decisions are borderline are excluded from the loss on both sides
    if border.any():
        dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
        g = ro.backward(o, dLn); dL = torch.from_numpy(dLn).to(dev)
    color.backward(dL)
    _grad_close("means3D", means3D.grad.cpu().numpy(), g["dL_dmeans3D"])
    _grad_close("means2D", means2D.grad.cpu().numpy(), g["dL_dmean2D"])
    _grad_close("opacity", opac.grad.cpu().numpy(), g["dL_dopacity"])
    _grad_close("sh", shs.grad.cpu().numpy(), g["dL_dsh"])
    _grad_close("scales", scales.grad.cpu().numpy(), g["dL_dscales"])
    _grad_close("rotations", rots.grad.cpu().numpy(), g["dL_drots"])


def test_backward_deterministic():
    """No float atomics: two backward runs give bitwise identical gradients."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(3000, 128, 96, 2, 4)
    rs = _settings(s, dev)
    outs = []
    for _ in range(2):
        t = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
        m, op, sh, sc, ro_ = t(s["means3D"]), t(s["opacities"]), t(s["shs"]), t(s["scales"]), t(s["rotations"])
        m2 = torch.zeros_like(m, requires_grad=True)
        color, _ = GaussianRasterizer(rs)(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=ro_)
        color.backward(torch.from_numpy(s["dL_dimage"]).to(dev))
        outs.append([x.grad.clone() for x in (m, m2, op, sh, sc, ro_)] + [color.detach().clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_colors_precomp_and_cov3d_precomp():
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(1500, 96, 96, 0, 8)
    rs_np = np.random.RandomState(0)
    colors = rs_np.uniform(0, 1, (1500, 3)).astype(np.float32)
    o = _oracle(s, colors=colors)
    g = ro.backward(o, s["dL_dimage"])
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    m, op, col, sc, rt = t(s["means3D"]), t(s["opacities"]), t(colors), t(s["scales"]), t(s["rotations"])
    m2 = torch.zeros_like(m, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=m, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=rt)
    border = o["margin"] < BORDER
    assert np.abs(color.detach().cpu().numpy() - o["color"]).max(0)[~border].max() <= RGB_TOL
    dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
    g = ro.backward(o, dLn)
    color.backward(torch.from_numpy(dLn).to(dev))
    _grad_close("colors", col.grad.cpu().numpy(), g["dL_dcolor"])
    _grad_close("means3D", m.grad.cpu().numpy(), g["dL_dmeans3D"])
    # cov3D_precomp path: feed the oracle's covariances
    cov = torch.from_numpy(o["cov3D"]).to(dev).requires_grad_(True)
    o2 = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], s["W"], s["H"],
                    s["tanfovx"], s["tanfovy"], s["bg"], colors_precomp=colors, cov3D_precomp=o["cov3D"])
    g2 = ro.backward(o2, dLn)
    m_, op_, col_ = t(s["means3D"]), t(s["opacities"]), t(colors)
    color2, radii2 = GaussianRasterizer(rs)(means3D=m_, means2D=torch.zeros_like(m_, requires_grad=True), opacities=op_,
                                            colors_precomp=col_, cov3D_precomp=cov)
    np.testing.assert_array_equal(radii2.cpu().numpy(), o2["radii"])
    color2.backward(torch.from_numpy(dLn).to(dev))
    _grad_close("cov3D", cov.grad.cpu().numpy(), g2["dL_dcov3D"])


def test_api_validation_and_edge_cases():
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(64, 40, 24, 1, 2)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    r = GaussianRasterizer(rs)
    m = t(s["means3D"])
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=t(s["opacities"]), scales=t(s["scales"]), rotations=t(s["rotations"]))
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=t(s["opacities"]), shs=t(s["shs"]))
    # all Gaussians behind the camera -> background image, radii 0
    behind = s["means3D"].copy(); behind[:, 2] = -1
    color, radii = r(means3D=t(behind), means2D=m, opacities=t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                     rotations=t(s["rotations"]))
    assert int(radii.abs().sum()) == 0
    assert torch.allclose(color, t(s["bg"])[:, None, None].expand_as(color))
    vis = r.markVisible(t(s["means3D"]))
    assert vis.dtype == torch.bool and vis.shape == (64,)
    assert torch.equal(vis.cpu(), torch.from_numpy(s["means3D"][:, 2] > 0.2))
    # CPU tensors are refused loudly (no fallback)
    with pytest.raises(RuntimeError):
        r(means3D=torch.from_numpy(s["means3D"]), means2D=m, opacities=t(s["opacities"]), shs=t(s["shs"]),
          scales=t(s["scales"]), rotations=t(s["rotations"]))


def test_config2_50k_512_forward():
    """BASELINE config 2: 50k Gaussians, 512x512, SH deg 0, forward; bit-exact binning."""
    from sings_amd.inspect_ws import forward_with_state
    dev = _dev()
    s = synthetic_scene(50000, 512, 512, 0, 2)
    o = _oracle(s)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                            rotations=t(s["rotations"]))
    _check_forward_state(s, st, o)
    _check_image(st["color"].cpu().numpy(), st["final_T"].cpu().numpy(), st["n_contrib"].cpu().numpy(), o)


@pytest.mark.parametrize("N,W,H", [(20000, 64, 64), (70000, 48, 32)])
def test_long_tile_lists_sort_paths(N, W, H):
    """Dense scenes: per-tile lists of thousands of entries exercise the workgroup sort (> 256 entries) and the
    chunk + rank path (> 4096 entries); binning must stay bit-exact and gradients correct."""
    from sings_amd.inspect_ws import forward_with_state
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(N, W, H, 1, 31)
    s["opacities"] = (s["opacities"] * 0.05).astype(np.float32)         # keep transmittance alive deep into the lists
    o = _oracle(s)
    tl = o["ranges"][:, 1].astype(int) - o["ranges"][:, 0]
    assert tl.max() > (4096 if N >= 70000 else 256)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                            rotations=t(s["rotations"]))
    _check_forward_state(s, st, o)
    border = o["margin"] < BORDER
    diff = np.abs(st["color"].cpu().numpy() - o["color"]).max(0)
    assert diff[~border].max() <= 2e-5          # thousands of blended terms per pixel: fp32 accumulation noise
    dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
    g = ro.backward(o, dLn)
    req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
    color, _ = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=op, shs=sh,
                                      scales=sc, rotations=rt)
    color.backward(torch.from_numpy(dLn).to(dev))
    # Lists longer than 256 entries are composited backward per depth segment, each segment starting from the forward
    # checkpoint: the colour behind the segment is (C_total - C_front) / T_front instead of the oracle's back-to-front
    # recursion -- the same quantity with a different fp32 rounding (~1e-6 per pixel term, summed over the tile).
    _grad_close("means3D", m.grad.cpu().numpy(), g["dL_dmeans3D"], rtol=1e-3, atol=6e-6)
    _grad_close("opacity", op.grad.cpu().numpy(), g["dL_dopacity"], rtol=1e-3, atol=6e-6)


def test_capacity_growth_and_empty_input():
    """The wrapper grows the pair workspace and re-runs when R exceeds it; P = 0 renders the background."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from sings_amd import rasterizer as rz
    dev = _dev()
    s = synthetic_scene(3000, 160, 128, 0, 9)
    s["scales"] = (s["scales"] * 30).astype(np.float32)                 # big splats: R exceeds the first-guess capacity
    o = _oracle(s)
    assert o["R"] > 4 * 3000 + 80 + (1 << 16)
    rz.reset_overflow_state()
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    color, radii = GaussianRasterizer(rs)(means3D=t(s["means3D"]), means2D=t(s["means3D"]), opacities=t(s["opacities"]),
                                          shs=t(s["shs"]), scales=t(s["scales"]), rotations=t(s["rotations"]))
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    strict = o["margin"] >= BORDER
    assert np.abs(color.cpu().numpy() - o["color"]).max(0)[strict].max() <= RGB_TOL
    # empty input
    e = torch.zeros((0, 3), device=dev)
    color, radii = GaussianRasterizer(rs)(means3D=e, means2D=e, opacities=torch.zeros((0, 1), device=dev),
                                          shs=torch.zeros((0, 16, 3), device=dev), scales=e,
                                          rotations=torch.zeros((0, 4), device=dev))
    assert radii.numel() == 0
    assert torch.allclose(color, t(s["bg"])[:, None, None].expand_as(color))


def test_many_tiles_packed_counters_vs_oracle():
    """> 4096 tiles: the binning switches to packed single counters per tile (the cfg3 regime).  Same bars as the small
    scenes: bit-exact binning, RGB <= 1e-5, gradients within tolerance."""
    from sings_amd.inspect_ws import forward_with_state
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(40000, 1600, 1056, 1, 41)
    assert ((s["W"] + 15) // 16) * ((s["H"] + 15) // 16) > 4096
    o = _oracle(s)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                            rotations=t(s["rotations"]))
    _check_forward_state(s, st, o)
    _check_image(st["color"].cpu().numpy(), st["final_T"].cpu().numpy(), st["n_contrib"].cpu().numpy(), o)
    border = o["margin"] < BORDER
    dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
    g = ro.backward(o, dLn)
    req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
    color, _ = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=op, shs=sh,
                                      scales=sc, rotations=rt)
    color.backward(torch.from_numpy(dLn).to(dev))
    for name, a, b in (("means3D", m.grad, g["dL_dmeans3D"]), ("opacity", op.grad, g["dL_dopacity"]),
                       ("scales", sc.grad, g["dL_dscales"]), ("rotations", rt.grad, g["dL_drots"]),
                       ("shs", sh.grad, g["dL_dsh"])):
        _grad_close(name, a.cpu().numpy().reshape(b.shape), b)


@pytest.mark.parametrize("opacity_scale", [0.25, 1.0])
def test_segmented_backward_with_early_termination(opacity_scale):
    """Tile lists of several 256-entry depth segments where pixels saturate INSIDE the list: segments behind the deepest
    contributor only write zero records, segments in front start from the forward checkpoints; every gradient is
    checked against the oracle's single back-to-front pass."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(24000, 96, 80, 2, 43)
    s["opacities"] = (s["opacities"] * opacity_scale).astype(np.float32)
    o = _oracle(s)
    tl = o["ranges"][:, 1].astype(int) - o["ranges"][:, 0]
    nc = o["n_contrib"]
    assert tl.max() > 3 * 256 and np.median(tl) > 256          # most tiles are segmented
    assert nc.max() < tl.max()                                  # and some pixels stop before the end of their list
    rs = _settings(s, dev)
    border = o["margin"] < BORDER
    dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
    g = ro.backward(o, dLn)
    req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
    m2 = torch.zeros_like(m, requires_grad=True)
    color, _ = GaussianRasterizer(rs)(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rt)
    diff = np.abs(color.detach().cpu().numpy() - o["color"]).max(0)
    assert diff[~border].max() <= 2e-5
    color.backward(torch.from_numpy(dLn).to(dev))
    for name, a, b in (("means3D", m.grad, g["dL_dmeans3D"]), ("means2D", m2.grad, g["dL_dmean2D"]),
                       ("opacity", op.grad, g["dL_dopacity"]), ("scales", sc.grad, g["dL_dscales"]),
                       ("rotations", rt.grad, g["dL_drots"]), ("shs", sh.grad, g["dL_dsh"])):
        _grad_close(name, a.cpu().numpy().reshape(b.shape), b, rtol=1e-3, atol=6e-6)


@pytest.mark.parametrize("N,W,H,seed", [(200000, 1920, 1080, 3), (500000, 2048, 2048, 5)])
def test_full_size_configs_against_the_oracle(N, W, H, seed):
    """BASELINE configs[2] ("200 k Gaussians, 1080p, SH deg 3, forward+backward, pixel-diff vs reference") and the raster part
    of configs[4] (500 k @ 2048 x 2048) AT FULL SIZE against the CPU oracle -- the operator call being replaced is
    gs_renderer_single.py:87-95.  The scalar C oracle needs ~4 s / ~10 s per view: the same bars as the small scenes --
    binning bit for bit (radii, rectangles, depth bits, keys, sorted lists, ranges, R), RGB <= 1e-5 off borderline pixels with
    the oracle's flip bound on the others, final_T / n_contrib, and EVERY gradient through the reference-facing autograd API."""
    from sings_amd.inspect_ws import forward_with_state
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(N, W, H, 3, seed)
    o = _oracle(s)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]),
                            rotations=t(s["rotations"]), capacity=5 * N)
    _check_forward_state(s, st, o)
    _check_image(st["color"].cpu().numpy(), st["final_T"].cpu().numpy(), st["n_contrib"].cpu().numpy(), o)
    del st
    border = o["margin"] < BORDER
    dLn = s["dL_dimage"].copy(); dLn[:, border] = 0
    g = ro.backward(o, dLn)
    req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
    m2 = torch.zeros_like(m, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rt)
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    assert np.abs(color.detach().cpu().numpy() - o["color"]).max(0)[~border].max() <= RGB_TOL
    color.backward(torch.from_numpy(dLn).to(dev))
    for name, a, b in (("means3D", m.grad, g["dL_dmeans3D"]), ("means2D", m2.grad, g["dL_dmean2D"]),
                       ("opacity", op.grad, g["dL_dopacity"]), ("scales", sc.grad, g["dL_dscales"]),
                       ("rotations", rt.grad, g["dL_drots"]), ("shs", sh.grad, g["dL_dsh"])):
        _grad_close(name, a.cpu().numpy().reshape(b.shape), b)


@pytest.mark.parametrize("N,W,H,seed", [(200000, 1920, 1080, 3), (500000, 2048, 2048, 5)])
def test_cfg3_full_size_properties(N, W, H, seed):
    """BASELINE configs[2] and configs[4] at full size (200 k Gaussians @ 1920x1080; 500 k @ 2048x2048 = 16 384 tiles, the
    packed-counter regime; SH degree 3) -- size-independent properties next to the oracle comparison above: every tile range is strictly sorted by (depth bits, Gaussian id), every
    Gaussian appears exactly once in each tile of its rectangle and nowhere else, R = sum of tiles touched, two runs are
    bitwise identical, and the backward pass is linear in dL/dimage."""
    from sings_amd.inspect_ws import forward_with_state
    from sings_amd.engine import RasterEngine
    dev = _dev()
    s = synthetic_scene(N, W, H, 3, seed)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    st = forward_with_state(rs, ins[0], ins[2], shs=ins[1], scales=ins[3], rotations=ins[4], capacity=5 * N)
    R = st["R"]
    gx = (W + 15) // 16
    radii = st["radii"].cpu().numpy(); rwh = st["rect_wh"].cpu().numpy(); rmin = st["rect_min"].cpu().numpy()
    tt = np.where(radii > 0, (rwh & 0xffff) * (rwh >> 16), 0).astype(np.int64)
    assert R == int(tt.sum())
    ranges = st["ranges"].cpu().numpy().astype(np.int64)
    n = ranges[:, 1] - ranges[:, 0]
    assert int(n.sum()) == R and (ranges[n > 0, 0] == np.concatenate([[0], np.cumsum(n[n > 0])[:-1]])).all()
    keys = st["point_keys"].cpu().numpy().astype(np.uint64); pl = st["point_list"].cpu().numpy().astype(np.int64)
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    assert (tile_of == np.repeat(np.arange(len(n)), n)).all()                    # ranges and keys agree
    depth_bits = st["depths"].cpu().numpy().view(np.uint32).astype(np.uint64)
    assert ((keys & np.uint64(0xffffffff)) == depth_bits[pl]).all()
    # strictly increasing (tile, depth bits, gid): compare neighbours inside each range
    same = tile_of[1:] == tile_of[:-1]
    k0, k1 = keys[:-1][same], keys[1:][same]
    g0, g1 = pl[:-1][same], pl[1:][same]
    assert ((k0 < k1) | ((k0 == k1) & (g0 < g1))).all()
    # membership: each pair's tile lies in the Gaussian's rectangle, and each Gaussian occurs tiles_touched times
    tx, ty = tile_of % gx, tile_of // gx
    x0, y0 = (rmin & 0xffff)[pl], (rmin >> 16)[pl]
    assert ((tx >= x0) & (tx < x0 + (rwh & 0xffff)[pl]) & (ty >= y0) & (ty < y0 + (rwh >> 16)[pl])).all()
    assert (np.bincount(pl, minlength=len(tt)) == tt).all()
    assert len(np.unique(tile_of * (1 << 20) + pl)) == R                          # no duplicate (tile, Gaussian)
    # determinism + linearity of the backward through the pre-allocated engine
    eng = RasterEngine(N, W, H, 16, dev, capacity_pairs=R + 4096)
    eng.set_camera(rs)
    rng = np.random.RandomState(0)
    d1 = t(s["dL_dimage"]); d2 = t(rng.standard_normal(s["dL_dimage"].shape).astype(np.float32))
    def grads(d):
        eng.forward(*ins); eng.backward(*ins, d)
        torch.cuda.synchronize()
        return eng.grad_flat.clone(), eng.color.clone()
    ga, ca = grads(d1); gb, cb = grads(d1)
    assert torch.equal(ga, gb) and torch.equal(ca, cb)
    assert torch.equal(ca, st["color"])
    g2, _ = grads(d2)
    g12, _ = grads(0.5 * d1 - 2.0 * d2)
    ref = 0.5 * ga.double() - 2.0 * g2.double()
    err = (g12.double() - ref).abs().max().item(); scale = ref.abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)
    # permutation invariance: the Gaussians in another order give the SAME image and, row for row, the same gradients -- bit for bit
    # (len(S) Gaussians share their depth bits with a tile neighbour)
    if True:
        # (equal depth bits inside a tile are ordered by Gaussian id: the permutation keeps the relative order of the few Gaussians
        #  that take part in such a tie, everything else moves freely)
        tie = k0 == k1
        S = np.unique(np.concatenate([g0[tie], g1[tie]]))
        new_of_old = np.argsort(np.random.RandomState(7).permutation(N))
        new_of_old[S] = np.sort(new_of_old[S])
        perm = torch.from_numpy(np.argsort(new_of_old)).to(dev)                  # perm[new] = old
        insp = [x[perm].contiguous() for x in ins]
        eng.forward(*insp); eng.backward(*insp, d1)
        torch.cuda.synchronize()
        assert torch.equal(eng.color, ca)
        o = 0
        for x in (ins[0], ins[3], ins[4], ins[2], ins[1]):        # grad_flat: means3D, scales, rotations, opacities, shs (engine.py)
            w = x[0].numel()
            a = ga[o:o + N * w].view(N, w); b = eng.grad_flat[o:o + N * w].view(N, w)
            assert torch.equal(b, a[perm]), "gradient rows follow the permutation"
            o += N * w


def test_capacity_overflow_is_reported_and_harmless():
    """R > capacity_pairs: the call reports R (so the caller can grow the workspace), never follows unwritten list
    entries (workspaces poisoned with 0xFF here) and renders the background."""
    import ctypes as C
    from sings_amd import _lib
    from sings_amd.rasterizer import _settings_struct, _ptr
    dev = _dev()
    s = synthetic_scene(3000, 160, 128, 0, 9)
    s["scales"] = (s["scales"] * 30).astype(np.float32)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    P, W, H, cap = 3000, s["W"], s["H"], 20000
    keep = []
    st = _settings_struct(rs, dev, 16, keep)
    L = _lib.layout(P, W, H, cap)
    poison = lambda n: torch.full((n,), 0xFF, dtype=torch.uint8, device=dev)
    geom, binning, img = poison(L.geom_bytes), poison(L.bin_bytes), poison(L.img_bytes)
    color = torch.empty((3, H, W), device=dev); radii = torch.empty(P, dtype=torch.int32, device=dev)
    nr = C.c_int64(0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.load().sg_rasterize_forward(
        C.byref(st), P, _ptr(ins[0]), _ptr(ins[1]), None, _ptr(ins[2]), _ptr(ins[3]), _ptr(ins[4]), None, _ptr(geom),
        _ptr(binning), cap, _ptr(img), _ptr(color), _ptr(radii), 0, C.byref(nr), stream), "forward")
    torch.cuda.synchronize()
    assert nr.value > cap
    bgv = torch.from_numpy(s["bg"]).to(dev)[:, None, None].expand(3, H, W)
    assert torch.equal(color, bgv)


def test_engine_step_replays_from_a_hip_graph():
    """A step of the pre-allocated engine neither synchronises nor allocates, so it can be captured into a HIP graph;
    the replay (after changing the inputs in place) gives exactly what direct calls give."""
    from sings_amd.engine import RasterEngine
    dev = _dev()
    s = synthetic_scene(5000, 256, 192, 2, 13)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(s["dL_dimage"])
    eng = RasterEngine(5000, s["W"], s["H"], 16, dev, capacity_pairs=8 * 5000 + 65536)
    eng.set_camera(rs)
    g = eng.capture(*ins, dL)
    ins[0].add_(0.01)                                           # in-place update: the graph reads the same buffers
    g.replay()
    torch.cuda.synchronize()
    color_g, grad_g = eng.color.clone(), eng.grad_flat.clone()
    eng.forward(*ins); eng.backward(*ins, dL)
    torch.cuda.synchronize()
    assert torch.equal(color_g, eng.color) and torch.equal(grad_g, eng.grad_flat)
    assert eng.num_rendered() <= eng.cap
    # replays separated by host synchronisations (a training loop that reads a loss value every step): with
    # hipMemsetAsync nodes in the graph every replay after the first synchronisation came out wrong on ROCm 7.2 --
    # the workspaces' counters are therefore zeroed by a kernel (sg_zero_async)
    R = eng.num_rendered()
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert eng.num_rendered() == R
        assert torch.equal(color_g, eng.color) and torch.equal(grad_g, eng.grad_flat)


def test_short_lists_hint_skips_the_long_sort_and_is_checked_on_the_device():
    """SG_FLAG_SHORT_LISTS (RasterEngine.set_camera(short_lists=True)): with every tile list <= 1024 entries (what the compositing workgroups
    sort themselves) the result is bit for bit the one without the hint (the long-list sort launches are simply not issued); when a longer list turns up
    nothing follows the unsorted list -- background image, zero work in backward -- and the pair count reads back as
    NUM_RENDERED_LONG_LIST."""
    from sings_amd import _lib
    from sings_amd.engine import RasterEngine
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s = synthetic_scene(6000, 256, 192, 2, 14)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(s["dL_dimage"])
    eng = RasterEngine(6000, s["W"], s["H"], 16, dev, capacity_pairs=8 * 6000 + 65536)
    eng.set_camera(_settings(s, dev))
    eng.forward(*ins); eng.backward(*ins, dL)
    R = eng.num_rendered()
    color0, grad0 = eng.color.clone(), eng.grad_flat.clone()
    n = eng.binning[eng.L.bin_ranges:eng.L.bin_ranges + 8 * 16 * 12].view(torch.int32).view(-1, 2)
    assert int((n[:, 1] - n[:, 0]).max()) <= 256
    eng.set_camera(_settings(s, dev), short_lists=True)
    eng.color.zero_(); eng.grad_flat.zero_()
    eng.forward(*ins); eng.backward(*ins, dL)
    assert eng.num_rendered() == R and torch.equal(eng.color, color0) and torch.equal(eng.grad_flat, grad0)
    # a scene with a long list: big splats piled on the image centre
    s2 = synthetic_scene(6000, 256, 192, 2, 14)
    s2["means3D"][:, :2] *= 0.05
    ins2 = [t(s2[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    eng.set_camera(_settings(s2, dev))
    eng.forward(*ins2)
    n = eng.binning[eng.L.bin_ranges:eng.L.bin_ranges + 8 * 16 * 12].view(torch.int32).view(-1, 2)
    assert int((n[:, 1] - n[:, 0]).max()) > 1024 and eng.num_rendered() > 0
    eng.set_camera(_settings(s2, dev), short_lists=True)
    eng.forward(*ins2); eng.backward(*ins2, dL)
    assert eng.num_rendered() == _lib.NUM_RENDERED_LONG_LIST
    bgv = t(s2["bg"])[:, None, None].expand_as(eng.color)
    assert torch.equal(eng.color, bgv)
    assert float(eng.d_means3D.abs().max()) == 0.0 and float(eng.d_sh.abs().max()) == 0.0


def test_direct_binning_leaves_the_bits_of_the_scatter_path():
    """Images of many tiles with SG_FLAG_SHORT_LISTS take DIRECT binning (include/sings_hip.h): the preprocess writes every pair's key into
    its tile's row of `tile_keys` at the rank its counting atomic returned -- no pair records, no scatter pass.  Ranges, sorted lists,
    image and every gradient carry the bits of the plain path, with one frame and with K frames per launch; a list longer than a row
    is refused exactly as the hint promises (background, no gradients, NUM_RENDERED_LONG_LIST)."""
    from sings_amd import _lib
    from sings_amd.engine import RasterEngine, RasterFramesEngine
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    N, W, H = 30000, 1280, 1040                                  # 80 x 65 = 5200 tiles: no per-workgroup histogram
    s = synthetic_scene(N, W, H, 3, 41)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(s["dL_dimage"])
    T = ((W + 15) // 16) * ((H + 15) // 16)
    eng = RasterEngine(N, W, H, 16, dev, capacity_pairs=16 * N + 65536)
    assert eng.L.bin_bytes - eng.L.bin_tile_keys >= T * 1024 * 8

    def lists(e):
        rg = e.binning[e.L.bin_ranges:e.L.bin_ranges + 8 * T].view(torch.int32).view(-1, 2).clone()
        R = int(rg[:, 1].max())
        return rg, e.binning[e.L.bin_point_list:e.L.bin_point_list + 4 * R].view(torch.int32).clone()

    eng.set_camera(_settings(s, dev))
    eng.forward(*ins); eng.backward(*ins, dL)
    R = eng.num_rendered()
    color0, grad0 = eng.color.clone(), eng.grad_flat.clone()
    rg0, pl0 = lists(eng)
    assert R > 3 * N and int((rg0[:, 1] - rg0[:, 0]).max()) <= 1024
    for rep in range(2):                                         # (twice: the counters the forward leaves zeroed serve the next call)
        eng.set_camera(_settings(s, dev), short_lists=True)
        eng.color.zero_(); eng.grad_flat.zero_()
        eng.forward(*ins); eng.backward(*ins, dL)
        rg1, pl1 = lists(eng)
        assert eng.num_rendered() == R and torch.equal(rg1, rg0) and torch.equal(pl1, pl0)
        assert torch.equal(eng.color, color0) and torch.equal(eng.grad_flat, grad0)
    # K = 2 frames per launch (two cameras), against two single-frame calls of the plain path
    sv = dict(s)
    view = s["viewmatrix"].copy(); view[3, 0] = 0.03
    sv["viewmatrix"] = view
    sv["projmatrix"] = (view @ (np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"])).astype(np.float32)
    sv["campos"] = np.linalg.inv(view)[3, :3].astype(np.float32)
    st0, st1 = _settings(s, dev), _settings(sv, dev)
    eng.set_camera(st1)
    eng.forward(*ins)
    color1 = eng.color.clone()
    fe = RasterFramesEngine(N, W, H, 16, 2, dev, capacity_pairs=16 * N + 65536)
    fe.set_camera(st0._replace(viewmatrix=torch.stack([st0.viewmatrix, st1.viewmatrix]), projmatrix=torch.stack([st0.projmatrix, st1.projmatrix]),
                               campos=torch.stack([st0.campos, st1.campos])), short_lists=True)
    fe.forward(*ins)
    assert torch.equal(fe.color[0], color0) and torch.equal(fe.color[1], color1)
    # a list longer than a row of tile_keys
    s2 = synthetic_scene(N, W, H, 3, 41)
    s2["means3D"][:, :2] *= 0.02
    ins2 = [t(s2[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    eng.set_camera(_settings(s2, dev))
    eng.forward(*ins2)
    rg2, _ = lists(eng)
    assert int((rg2[:, 1] - rg2[:, 0]).max()) > 1024 and eng.num_rendered() > 0
    color2 = eng.color.clone()
    eng.set_camera(_settings(s2, dev), short_lists=True)
    eng.forward(*ins2); eng.backward(*ins2, dL)
    assert eng.num_rendered() == _lib.NUM_RENDERED_LONG_LIST
    assert torch.equal(eng.color, t(s2["bg"])[:, None, None].expand_as(eng.color)) and float(eng.d_means3D.abs().max()) == 0.0
    eng.set_camera(_settings(s2, dev))                           # ... and the plain path still renders it (nothing stale left behind)
    eng.forward(*ins2)
    assert torch.equal(eng.color, color2)


def test_drop_in_surface_learns_the_short_list_hint_and_never_returns_a_wrong_frame():
    """GaussianRasterizer in its default ("sync") mode: the count word of a frame says whether every list had <= 512 entries; if so the
    NEXT frame of that (device, P, W, H) runs with SG_FLAG_SHORT_LISTS (direct binning at this image size) and gives the same bits.  When
    the scene then grows a long list under the hint, the frame is rendered again without it before the call returns."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from sings_amd import rasterizer as rz
    dev = _dev()
    N, W, H = 30000, 1280, 1040
    rz.reset_overflow_state(dev)
    key = (dev.index, N, W, H)

    def frame(sc):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(True)
        ins = dict(means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
        m2d = torch.zeros_like(ins["means3D"], requires_grad=True)
        color, _ = GaussianRasterizer(_settings(sc, dev))(means2D=m2d, **ins)
        color.backward(torch.from_numpy(sc["dL_dimage"]).to(dev))
        return color.detach().clone(), [ins[k].grad.clone() for k in ("means3D", "opacities", "shs", "scales", "rotations")] + [m2d.grad.clone()]

    s = synthetic_scene(N, W, H, 3, 41)
    c0, g0 = frame(s)                                            # plain path; learns the hint
    assert rz._short_ok.get(key) is True
    c1, g1 = frame(s)                                            # with the hint
    assert torch.equal(c1, c0) and all(torch.equal(a, b) for a, b in zip(g1, g0))
    s2 = synthetic_scene(N, W, H, 3, 41)
    s2["means3D"][:, :2] *= 0.02                                 # lists of more than 1024 entries
    c2, g2 = frame(s2)                                           # under the hint: refused on the device, rendered again without it
    assert rz._short_ok.get(key) is False
    rz.reset_overflow_state(dev)
    c3, g3 = frame(s2)                                           # never hinted
    assert torch.equal(c2, c3) and all(torch.equal(a, b) for a, b in zip(g2, g3))
    assert float((c2 - torch.from_numpy(s2["bg"]).to(dev)[:, None, None]).abs().max()) > 0.1
    rz.reset_overflow_state(dev)


def test_views_in_flight_on_two_streams_match_sequential_runs():
    """bench.py renders the views of a step through one engine per view (own workspaces, own row of a shared gradient
    buffer) spread over two HIP streams.  The library keeps no state between calls, so views in flight at the same time
    must give bit for bit what the same engines give one after the other on one stream."""
    from sings_amd.engine import RasterEngine
    dev = _dev()
    s = synthetic_scene(20000, 512, 384, 3, 21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(s["dL_dimage"])
    K = 4
    per = 20000 * (3 + 3 + 4 + 1 + 3 * 16)
    grads = torch.zeros((K, per), device=dev)
    engs = []
    for v in range(K):
        sv = dict(s)
        view = s["viewmatrix"].copy(); view[3, 0] = 0.05 * v
        sv["viewmatrix"] = view
        sv["projmatrix"] = (view @ (np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"])).astype(np.float32)
        sv["campos"] = np.linalg.inv(view)[3, :3].astype(np.float32)
        e = RasterEngine(20000, s["W"], s["H"], 16, dev, capacity_pairs=8 * 20000 + 65536, grad_flat=grads[v])
        e.set_camera(_settings(sv, dev))
        engs.append(e)
    for e in engs:                                               # sequential reference
        e.forward(*ins); e.backward(*ins, dL)
    torch.cuda.synchronize()
    ref_g = grads.clone(); ref_c = [e.color.clone() for e in engs]
    assert not torch.equal(ref_c[0], ref_c[1])                   # (the cameras really differ)
    grads.zero_()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    cur = torch.cuda.current_stream(dev)
    for rep in range(3):
        for st in streams:
            st.wait_stream(cur)
        for v, e in enumerate(engs):
            with torch.cuda.stream(streams[v % 2]):
                e.forward(*ins); e.backward(*ins, dL)
        for st in streams:
            cur.wait_stream(st)
    total = torch.sum(grads, dim=0)
    torch.cuda.synchronize()
    assert torch.equal(grads, ref_g)
    # the same through the library's helper
    from sings_amd.engine import ViewBatch
    grads.zero_()
    acc = ViewBatch(engs, grads, streams=3).run(lambda v, e: (e.forward(*ins), e.backward(*ins, dL)))
    torch.cuda.synchronize()
    assert torch.equal(grads, ref_g) and torch.equal(acc, total)
    for e, c in zip(engs, ref_c):
        assert torch.equal(e.color, c)
    assert torch.equal(total, torch.sum(ref_g, dim=0))
    with pytest.raises(ValueError):
        RasterEngine(20000, s["W"], s["H"], 16, dev, capacity_pairs=1024, grad_flat=torch.zeros(7, device=dev))


def test_views_of_a_step_add_to_one_gradient_buffer_in_a_fixed_order():
    """Round 3: the views of a step share ONE gradient buffer (ViewBatch `chain`): view 0's per-Gaussian backward writes it,
    every later view ADDS to it (sg_rasterize_backward_gaussians(accumulate=1)) after the event of the view in front of it.
    The result must be, bit for bit, ((g0 + g1) + g2) + g3 of the per-view gradients -- whatever the number of streams, run
    after run -- and dL/dmeans2D stays per view."""
    from sings_amd.engine import RasterEngine, ViewBatch
    dev = _dev()
    s = synthetic_scene(20000, 512, 384, 3, 21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(s["dL_dimage"])
    K = 4
    per = 20000 * (3 + 3 + 4 + 1 + 3 * 16)

    def engines(buffers):
        engs = []
        for v in range(K):
            sv = dict(s)
            view = s["viewmatrix"].copy(); view[3, 0] = 0.05 * v
            sv["viewmatrix"] = view
            sv["projmatrix"] = (view @ (np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"])).astype(np.float32)
            sv["campos"] = np.linalg.inv(view)[3, :3].astype(np.float32)
            e = RasterEngine(20000, s["W"], s["H"], 16, dev, capacity_pairs=8 * 20000 + 65536, grad_flat=buffers[v])
            e.set_camera(_settings(sv, dev))
            engs.append(e)
        return engs
    rows = torch.zeros((K, per), device=dev)
    per_view = engines([rows[v] for v in range(K)])
    for e in per_view:
        e.forward(*ins); e.backward(*ins, dL)
    torch.cuda.synchronize()
    want = rows[0].clone()
    for v in range(1, K):
        want += rows[v]                                              # ((g0 + g1) + g2) + g3: one rounding per addition, as the kernel
    m2d = [e.d_means2D.clone() for e in per_view]
    acc = torch.full((per,), float("nan"), device=dev)               # (never pre-zeroed: view 0 WRITES)
    shared = engines([acc] * K)
    for n_streams in (1, 2, 3, 3):
        acc.fill_(float("nan"))
        batch = ViewBatch(shared, acc, streams=n_streams)
        assert batch.chain == (n_streams > 1)
        out = batch.run(lambda v, e: (e.forward(*ins), e.backward(*ins, dL)))
        torch.cuda.synchronize()
        assert out.data_ptr() == acc.data_ptr() and torch.equal(acc, want), n_streams
        for e, m in zip(shared, m2d):
            assert torch.equal(e.d_means2D, m)
    with pytest.raises(ValueError):
        ViewBatch(per_view, acc, streams=2)                          # views of one row must share the buffer
    # one row per STREAM (bench.py's default): views 0, 2 -> row 0, views 1, 3 -> row 1, each row in view order, then the fold
    two = torch.full((2, per), float("nan"), device=dev)
    by_stream = engines([two[v % 2] for v in range(K)])
    for rep in range(2):
        two.fill_(float("nan"))
        batch = ViewBatch(by_stream, two, streams=2)
        assert not batch.chain and batch.rows == 2
        out = batch.run(lambda v, e: (e.forward(*ins), e.backward(*ins, dL)))
        torch.cuda.synchronize()
        assert torch.equal(two[0], rows[0] + rows[2]) and torch.equal(two[1], rows[1] + rows[3])
        assert torch.equal(out, torch.sum(two, dim=0))


def test_against_frozen_oracle_vectors():
    """The HIP path through the drop-in autograd surface against the committed fixture tests/golden/raster_golden.npz (G6:
    frozen outputs of the restatement): radii / rectangles / sorted lists / tile ranges bit for bit, RGB within 1e-5 away
    from the hard thresholds, gradients within the parity tolerance."""
    import os
    from sings_amd.inspect_ws import forward_with_state
    from sings_amd.rasterizer import GaussianRasterizer
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_golden.npz"))
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for tag in ("a", "b", "c"):
        N, W, H, deg, seed = (int(v) for v in G[f"{tag}_case"])
        s = synthetic_scene(N, W, H, deg, seed)
        rs = _settings(s, dev)
        st = forward_with_state(rs, t(s["means3D"]), t(s["opacities"]), shs=t(s["shs"]), scales=t(s["scales"]), rotations=t(s["rotations"]))
        assert st["R"] == int(G[f"{tag}_R"])
        np.testing.assert_array_equal(st["radii"].cpu().numpy(), G[f"{tag}_radii"])
        np.testing.assert_array_equal(st["point_list"].cpu().numpy(), G[f"{tag}_point_list"])
        np.testing.assert_array_equal(st["ranges"].cpu().numpy(), G[f"{tag}_ranges"])
        strict = G[f"{tag}_margin"] >= 2e-5
        err = np.abs(st["color"].cpu().numpy() - G[f"{tag}_color"]).max(0)
        assert err[strict].max() <= RGB_TOL, (tag, err[strict].max())
        req = lambda a: t(a).requires_grad_(True)
        m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
        color, _ = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=op, shs=sh,
                                          scales=sc, rotations=rt)
        color.backward(t(G[f"{tag}_dL"]))
        for name, a in (("dL_dmeans3D", m.grad), ("dL_dscales", sc.grad), ("dL_drots", rt.grad), ("dL_dopacity", op.grad),
                        ("dL_dsh", sh.grad)):
            b = G[f"{tag}_{name}"].astype(np.float64); a = a.cpu().numpy().astype(np.float64)
            assert (np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-6 * np.abs(b).max()).all(), (tag, name, np.abs(a - b).max())


def test_deferred_overflow_check():
    """Deferred mode: the forward never reads the pair count (no host synchronisation inside a step).  A call whose count
    exceeds the capacity renders the background; check_deferred_overflow() reports it once, grows the capacity, and the
    next call is complete and equal to the synchronous result."""
    from sings_amd import rasterizer as rz
    from sings_amd.rasterizer import GaussianRasterizer
    dev = _dev()
    s = synthetic_scene(4000, 160, 128, 1, 29)
    rs = _settings(s, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = dict(means3D=t(s["means3D"]), means2D=torch.zeros(4000, 3, device=dev), opacities=t(s["opacities"]), shs=t(s["shs"]),
                scales=t(s["scales"]), rotations=t(s["rotations"]))
    ref, _ = GaussianRasterizer(rs)(**args)                      # synchronous mode (default)
    hint = dict(rz._capacity_hint)
    try:
        rz.set_deferred_overflow_check(True, capacity_pairs=1, device=dev)
        rz._capacity_hint[dev.index] = 1
        # (the wrapper never goes below max(hint, 4 P + tiles, 2^16) pairs: make the scene exceed that floor instead)
        big = synthetic_scene(30000, 160, 128, 1, 31)
        big["scales"] = big["scales"] * 6.0
        bargs = dict(means3D=t(big["means3D"]), means2D=torch.zeros(30000, 3, device=dev), opacities=t(big["opacities"]),
                     shs=t(big["shs"]), scales=t(big["scales"]), rotations=t(big["rotations"]))
        brs = _settings(big, dev)
        img, _ = GaussianRasterizer(brs)(**bargs)
        bgv = t(big["bg"])[:, None, None].expand(3, big["H"], big["W"])
        with pytest.raises(RuntimeError, match="deferred forward produced"):
            rz.check_deferred_overflow(dev)
        assert torch.equal(img, bgv)                             # overflow: background, nothing followed stale lists
        img2, _ = GaussianRasterizer(brs)(**bargs)               # capacity has grown
        R = rz.check_deferred_overflow(dev)
        assert R is not None and R > 4 * 30000 + 80 and not torch.equal(img2, bgv)
        rz.set_deferred_overflow_check(False)
        img3, _ = GaussianRasterizer(brs)(**bargs)
        assert torch.equal(img2, img3)
        out, _ = GaussianRasterizer(rs)(**args)
        assert torch.equal(out, ref)
    finally:
        rz.set_deferred_overflow_check(False)
        rz.reset_overflow_state()
        rz._capacity_hint.update(hint)


def test_default_overflow_mode_never_returns_a_wrong_frame():
    """The drop-in default ("sync", as upstream: R is known before the forward returns): a frame with > 2x the pairs of
    anything seen before is re-run with a larger workspace inside the SAME call -- never the background, never zero
    gradients.  The count comes back through the early-count word (SgRasterSettings.count_signal), i.e. without a stream
    synchronisation: the composite kernel may still be running when the call returns; results are complete once the
    stream is."""
    from sings_amd import rasterizer as rz
    from sings_amd.rasterizer import GaussianRasterizer
    dev = _dev()
    assert rz._mode == {"mode": "sync", "on_overflow": "raise"}
    rz.reset_overflow_state()
    big = synthetic_scene(30000, 160, 128, 1, 31)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    def args(scale):
        return dict(means3D=t(big["means3D"]), means2D=torch.zeros(30000, 3, device=dev), opacities=t(big["opacities"]),
                    shs=t(big["shs"]), scales=t(big["scales"] * scale).requires_grad_(True), rotations=t(big["rotations"]))
    brs = _settings(big, dev)
    bgv = t(big["bg"])[:, None, None].expand(3, big["H"], big["W"])
    try:
        small, _ = GaussianRasterizer(brs)(**args(0.3))
        cap0 = rz._capacity_hint[dev.index]
        assert rz._signal.get(dev.index) is not None and not rz._pending.get(dev.index)   # early-count words in use, nothing async
        a8 = args(8.0)
        huge, _ = GaussianRasterizer(brs)(**a8)                     # > 2x the pairs of the frame before
        assert rz._capacity_hint[dev.index] > cap0
        assert not torch.equal(huge, bgv)
        huge.sum().backward()
        assert float(a8["scales"].grad.abs().sum()) > 0
        again, _ = GaussianRasterizer(brs)(**args(8.0))             # same frame, now with a capacity that fits from the start
        assert torch.equal(huge, again)
        back, _ = GaussianRasterizer(brs)(**args(0.3))
        assert torch.equal(back, small)
        # the same through the synchronous read (debug = True syncs after every kernel and does not arm the word)
        dbg = brs._replace(debug=True)
        rz.reset_overflow_state()
        GaussianRasterizer(dbg)(**args(0.3))
        huge_d, _ = GaussianRasterizer(dbg)(**args(8.0))
        assert torch.equal(huge_d, huge)
    finally:
        rz.reset_overflow_state()


def test_async_overflow_check_is_opt_in_and_raises_late():
    """Opt-in "async" mode: the FIRST forward of a (device, P, image size) reads the pair count before returning and sizes
    the capacity with 2x headroom; later forwards copy (R, flag) to pinned memory behind their kernels and return at once.
    Several forwards may be issued before anything is looked at and none of their results is lost: a frame that overflowed
    renders the background, and the NEXT call RAISES (on_overflow="raise" is the default; "warn" downgrades it to a
    RuntimeWarning) with a capacity that fits from then on."""
    import warnings
    from sings_amd import rasterizer as rz
    from sings_amd.rasterizer import GaussianRasterizer
    dev = _dev()
    rz.reset_overflow_state()
    big = synthetic_scene(30000, 160, 128, 1, 31)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    def args(scale):
        return dict(means3D=t(big["means3D"]), means2D=torch.zeros(30000, 3, device=dev), opacities=t(big["opacities"]),
                    shs=t(big["shs"]), scales=t(big["scales"] * scale), rotations=t(big["rotations"]))
    brs = _settings(big, dev)
    bgv = t(big["bg"])[:, None, None].expand(3, big["H"], big["W"])
    try:
        rz.set_overflow_check("async")
        assert rz._mode["on_overflow"] == "raise"
        small, _ = GaussianRasterizer(brs)(**args(0.3))              # first call of this signature: synchronous
        assert not rz._pending.get(dev.index)
        cap0 = rz._capacity_hint[dev.index]
        small2, _ = GaussianRasterizer(brs)(**args(0.3))             # asynchronous from now on
        assert len(rz._pending[dev.index]) == 1 and torch.equal(small, small2)
        huge, _ = GaussianRasterizer(brs)(**args(8.0))               # > 2x the pairs: overflows the capacity, not yet known
        assert torch.equal(huge, bgv)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="rendered the background"):     # late, but loud
            GaussianRasterizer(brs)(**args(0.3))
        assert rz._capacity_hint[dev.index] > cap0
        redo, _ = GaussianRasterizer(brs)(**args(8.0))               # capacity grown by now
        assert not torch.equal(redo, bgv)
        # "warn": the same event as a RuntimeWarning
        rz.reset_overflow_state()
        rz.set_overflow_check("async", on_overflow="warn")
        GaussianRasterizer(brs)(**args(0.3)); GaussianRasterizer(brs)(**args(0.3))
        huge, _ = GaussianRasterizer(brs)(**args(8.0))
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            also, _ = GaussianRasterizer(brs)(**args(0.3))
        assert sum("rendered the background" in str(w.message) for w in rec) == 1 and torch.equal(also, small)
        rz.set_overflow_check("sync", on_overflow="raise")
        ref, _ = GaussianRasterizer(brs)(**args(8.0))
        assert torch.equal(redo, ref)
        assert rz.check_deferred_overflow(dev) is not None           # drains what is pending; nothing overflowed since
    finally:
        rz.set_overflow_check("sync", on_overflow="raise")
        rz.reset_overflow_state()


@pytest.mark.parametrize("n", [128, 129, 255, 256, 257, 512, 513, 1024, 1025, 4096, 4097, 12288, 12289])
def test_list_lengths_on_internal_boundaries(n):
    """A tile whose list has at least n entries, n around the boundaries of the rank sort (128: SG_RANKSORT_MAX), the depth
    segments (256), the in-composite sort (1024: SG_WSORT_MAX; beyond it the bucket sort, and on this few-tile image the
    four-workgroup composite and the sparse backward) and of the LDS-resident bucket sort (12 288): binning bit-exact, image and
    gradients vs the oracle (the sweep of tests/tools/fuzz_parity.py)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    R, mx = fz.check(fz.crafted(n, n), f"crafted {n}")
    assert mx >= n


def test_bucket_sort_of_clustered_depths():
    """Long lists are bucket-sorted by key value (sg_tile_partition_kernel + sg_group_sort_kernel).  Depth clusters exercise what
    uniform depths never reach: a bucket of > 1024 keys split again over its own key range (identical depth bits -> by Gaussian
    id) and the counting order for > 1024 keys on one float code next to another code.  Bit-exact sorted list, image, gradients."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    z = fz.clustered_depths(7)
    R, mx = fz.check(fz.crafted(z.size, 7, depths=z), "clustered depths")
    assert mx >= z.size



def test_throughput_flag_changes_the_schedule_not_the_result():
    """SG_FLAG_THROUGHPUT (set by ViewBatch when several views share the GPU) turns the four-workgroup composite of long tile
    lists off: same pixels, same entries, same order -- image, pair count and every gradient must be bit-identical, on a few-tile
    frame whose hot tile has a list of more than 3000 entries."""
    import importlib.util, os
    from sings_amd.engine import RasterEngine
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    dev = _dev()
    s = fz.crafted(3000, 21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = s["means3D"].shape[0]
    args = [t(s["means3D"]), t(s["shs"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])]
    dL = t(s["dL_dimage"])
    out = []
    for flag in (False, True):
        eng = RasterEngine(P, s["W"], s["H"], s["shs"].shape[1], dev, capacity_pairs=8 * P + 4096)
        eng.set_camera(fz.settings(s))
        eng.throughput = flag
        R = eng.forward(*args, sync_num_rendered=True)
        eng.backward(*args, dL)
        torch.cuda.synchronize()
        out.append((R, eng.color.clone(), eng.grad_flat.clone(), eng.d_means2D.clone()))
    assert out[0][0] == out[1][0] and out[0][0] > 3000
    for a, b in zip(out[0][1:], out[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_against_the_reference_spherical_harmonics(deg):
    """Golden G9 on the HIP path: the colour sg_preprocess_fwd_kernel<D> stores for a Gaussian (record words r, g, b; SH clamp
    bits) against the REFERENCE's eval_sh (sings/rec/utils/visualize/spherical_harmonics.py:54-113, executed in the build
    container by tests/golden/gen_sh_golden.py) + 0.5, clamped at 0 -- 512 directions, every octant, degrees 0..3; and
    dL/dsh of a full backward = reference basis value x the oracle's dL/dcolour."""
    import sh_case
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sings_amd.inspect_ws import forward_with_state
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seen = 0
    for g in sh_case.groups():
        rs = GaussianRasterizationSettings(image_height=sh_case.H, image_width=sh_case.W, tanfovx=sh_case.TANFOV, tanfovy=sh_case.TANFOV,
                                           bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=t(g["viewmatrix"]),
                                           projmatrix=t(g["projmatrix"]), sh_degree=deg, campos=t(g["campos"]), prefiltered=False,
                                           debug=False)
        st = forward_with_state(rs, t(g["means3D"]), t(g["opacities"]), shs=t(g["shs"]), scales=t(g["scales"]), rotations=t(g["rotations"]))
        assert bool((st["radii"] > 0).all())
        rgb, clamped = sh_case.expected_rgb(deg, g["idx"])
        got = st["rgb"].cpu().numpy()
        assert np.abs(got - rgb).max() <= 2e-6, np.abs(got - rgb).max()
        sure = np.abs(sh_case.G9[f"eval_deg{deg}"][g["idx"]] + 0.5) > 1e-5
        bits = st["clamp_bits"].cpu().numpy()
        got_clamped = np.stack([(bits >> c) & 1 for c in range(3)], 1).astype(bool)
        assert np.array_equal(got_clamped[sure], clamped[sure])
        seen += g["idx"].size
        if deg == 3:
            o = ro.forward(g["means3D"], g["opacities"], g["viewmatrix"], g["projmatrix"], g["campos"], sh_case.W, sh_case.H,
                           sh_case.TANFOV, sh_case.TANFOV, np.zeros(3, np.float32), scales=g["scales"], rotations=g["rotations"],
                           shs=g["shs"], sh_degree=deg)
            dL = np.random.RandomState(1).normal(0, 1, (3, sh_case.H, sh_case.W)).astype(np.float32)
            dL[:, o["margin"] < BORDER] = 0
            gr = ro.backward(o, dL)
            dcol = np.where(o["clamped"].astype(bool), 0.0, gr["dL_dcolor"])
            want = sh_case.G9["basis_deg3"][g["idx"]][:, :, None] * dcol[:, None, :]
            sh = t(g["shs"]).requires_grad_(True)
            m = t(g["means3D"])
            color, _ = GaussianRasterizer(rs)(means3D=m, means2D=torch.zeros_like(m), opacities=t(g["opacities"]), shs=sh,
                                              scales=t(g["scales"]), rotations=t(g["rotations"]))
            color.backward(t(dL))
            _grad_close("dL/dsh vs reference basis", sh.grad.cpu().numpy(), want, rtol=2e-4, atol=2e-6)
    assert seen == 512
