"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, workspace layout arithmetic, and the Python surface mirrors diff_gaussian_rasterization."""
import inspect
import os
import re

import pytest
import torch


def test_library_exports_every_declared_symbol(repo_root):
    from sings_amd import _lib
    hdr = open(os.path.join(repo_root, "include", "sings_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sg_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sings_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)
    assert b"gfx950" in lib.sg_version()


def test_layout_is_consistent():
    from sings_amd import _lib
    L = _lib.layout(1000, 1920, 1080, 50000)
    T = 120 * 68
    assert L.geom_recB - L.geom_recA == 16 and L.geom_recC - L.geom_recA == 32 and L.geom_bytes >= 1000 * 72   # 64-B records + depth + flags
    assert L.bin_ranges - L.bin_tile_count >= T * 4                   # (counters in 4x4 blocks of tiles: 120 x 68 tiles need no padding)
    assert L.bin_point_list - L.bin_pair_keys >= 50000 * 8
    assert L.img_n_contrib - L.img_final_T >= 1920 * 1080 * 4
    assert 50000 * 36 <= L.bwd_bytes < 50000 * 40
    for f, _ in L._fields_:
        assert getattr(L, f) % 256 == 0 or f in ("geom_recB", "geom_recC")     # (vectors 1 and 2 of the interleaved 64-B records)
    with pytest.raises(RuntimeError):
        _lib.layout(10, 0, 10, 10)
    # work items pack the tile id into 20 bits: 2^20 tiles or more are rejected, one tile less is accepted
    _lib.layout(10, 16 * 1023, 16 * 1025, 10)                      # 1023 * 1025 = 2^20 - 1 tiles
    with pytest.raises(RuntimeError, match="2\\^20 tiles"):
        _lib.layout(10, 16 * 1024, 16 * 1024, 10)
    # ... and by the entry points themselves (validated before anything is launched or dereferenced)
    import ctypes as C
    lib = _lib.load()
    st = _lib.SgRasterSettings()
    st.image_height = st.image_width = 16 * 1024
    st.tanfovx = st.tanfovy = 0.5; st.scale_modifier = 1.0; st.sh_degree = 0; st.sh_coeffs = 1
    st.bg = st.viewmatrix = st.projmatrix = st.campos = 0x1000                   # never dereferenced
    fake = C.c_void_p(0x1000)
    rc = lib.sg_rasterize_forward(C.byref(st), 1, fake, fake, None, fake, fake, fake, None, fake, fake, 10, fake, fake, fake, 0,
                                  None, None)
    assert rc != 0 and b"bad settings" in lib.sg_last_error()


def test_python_surface_matches_upstream_package():
    import diff_gaussian_rasterization as d
    fields = d.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(d.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                         "rotations", "cov3D_precomp"]
    assert list(inspect.signature(d.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp",
        "raster_settings"]
    assert issubclass(d._RasterizeGaussians, torch.autograd.Function)
    assert hasattr(d.GaussianRasterizer, "markVisible")


def test_argument_validation_and_no_cpu_fallback():
    import diff_gaussian_rasterization as d
    rs = d.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                         torch.zeros(3), False, False)
    r = d.GaussianRasterizer(rs)
    m = torch.zeros(4, 3); o = torch.ones(4, 1); sh = torch.zeros(4, 16, 3); sc = torch.ones(4, 3); q = torch.zeros(4, 4)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, scales=sc, rotations=q)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, shs=sh, colors_precomp=m, scales=sc, rotations=q)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh, scales=sc, rotations=q, cov3D_precomp=torch.zeros(4, 6))
    # CPU tensors: the product path refuses instead of silently falling back
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(m, m, o, shs=sh, scales=sc, rotations=q)


def test_bench_algorithmic_bytes_follow_the_survey_formula():
    """bench.py's roofline numerators are SURVEY.md 8(d): B = N (2 in + gout + 4 + 2 rec) + HW 40 + R 20, in = 44 + 12 (deg+1)^2,
    rec = 75, gout = 248; cfg3 with R = 8.4e5 is the survey's 0.27 GB/view."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    per, total = bench.algorithmic_bytes(200000, 1080, 1920, 840000, 3)
    assert total == 200000 * (2 * 236 + 248 + 4 + 150) + 2073600 * 40 + 840000 * 20
    assert abs(total / 1e9 - 0.2745) < 1e-3
    assert per["sg_render_bwd_kernel"] == 2073600 * 20 + 840000 * 4
    assert sum(per.values()) <= total + 840000 * 12       # the split never exceeds the whole-pass figure (+ binning keys)
