"""Host-side logic of bench.py that needs no GPU: the PMC sidecar guard (a committed counter pass is only used for the
roofline if it profiled THIS configuration and THIS tree), the repeated timed region, the per-kernel byte split."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CFG = {"workload": "raster", "gaussians": 200000, "width": 1920, "height": 1080, "sh_degree": 3}


def _profiles(tmp_path, meta, traffic_meta="same"):
    p = tmp_path / "profiles"
    p.mkdir()
    (p / "r99_pmc_SQ.csv").write_text("kernel,Counter_Name,mean,count\n"
                                      "sg_render_bwd_kernel,SQ_INSTS_VALU,80000000.0,5\n"
                                      "sg_preprocess_bwd_kernel<3>,SQ_INSTS_VALU,6000000.0,5\n")
    if meta is not None:
        (p / "r99_pmc_SQ.meta.json").write_text(json.dumps(meta))
    t = {"sg_render_bwd_kernel": 123456, "sg_preprocess_bwd_kernel<3>": 777}
    if traffic_meta is not None:
        t["_meta"] = meta if traffic_meta == "same" else traffic_meta
    (p / "hbm_traffic.json").write_text(json.dumps(t))
    return str(p)


def test_matching_sidecar_is_used(tmp_path):
    meta = {"config": dict(CFG), "sources": bench.source_hashes()}
    r = bench._committed_pmc("sg_render_bwd_kernel", CFG, _profiles(tmp_path, meta))
    assert r["stale"] is None and r["valu"] == 8.0e7 and r["traffic"] == 123456
    # template instances are matched by their base name
    (tmp_path / "b").mkdir()
    assert bench._committed_pmc("sg_preprocess_bwd_kernel", CFG, _profiles(tmp_path / "b", meta))["valu"] == 6.0e6


def test_tampered_sidecar_drops_the_valu_roofline(tmp_path):
    """One changed kernel source (here: a hash that is not the tree's) and the committed instruction count is not used:
    build_roofline falls back to the HBM roofline and says why."""
    src = bench.source_hashes()
    src["sings_amd/csrc/sg_render.hip"] = "0" * 40
    meta = {"config": dict(CFG), "sources": src}
    pdir = _profiles(tmp_path, meta)
    r = bench._committed_pmc("sg_render_bwd_kernel", CFG, pdir)
    assert "valu" not in r and "sg_render.hip" in r["stale"] and "traffic" not in r and "sg_render.hip" in r["traffic_stale"]
    R = bench._roofline                                          # (build_roofline looks the helpers up in its own module)
    old = R._committed_pmc
    R._committed_pmc = lambda k, c: old(k, c, pdir)
    try:
        kern = {"sg_preprocess_fwd_kernel": 0.03, "sg_render_fwd_kernel": 0.07, "sg_render_bwd_kernel": 0.15,
                "sg_preprocess_bwd_kernel": 0.03}
        per, total = bench.algorithmic_bytes(200000, 1080, 1920, 780000, 3)
        roof, valu = bench.build_roofline(kern, per, CFG, total, 0.25e-3, 6200.0)
    finally:
        R._committed_pmc = old
    assert roof["bound"] == "hbm" and roof["dominant_kernel"] == "sg_render_bwd_kernel"
    assert valu["frac"] is None and "sg_render.hip" in valu["note"]
    assert roof["dominant_kernel_traffic"] is None and roof["peak"] == 6200.0
    assert abs(roof["achieved"] - total / 0.25e-3 / 1e9) < 1e-6 and abs(roof["frac"] - roof["achieved"] / 6200.0) < 1e-12
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(roof)


def test_roofline_is_the_whole_pass_hbm_quantity_and_valu_is_secondary(tmp_path):
    """`roofline` = algorithmic bytes per view / seconds per view against the measured copy bandwidth;
    the VALU-issue figure of the dominant composite kernel is `roofline_valu`; `traffic` = the PMC bytes of ALL kernels."""
    meta = {"config": dict(CFG), "sources": bench.source_hashes()}
    pdir = _profiles(tmp_path, meta)
    R = bench._roofline
    old, oldt = R._committed_pmc, R.pmc_view_traffic
    R._committed_pmc = lambda k, c: old(k, c, pdir)
    R.pmc_view_traffic = lambda c, frames=1: oldt(c, pdir, frames=frames)
    try:
        kern = {"sg_preprocess_fwd_kernel": 0.03, "sg_render_fwd_kernel": 0.07, "sg_render_bwd_kernel": 0.15,
                "sg_preprocess_bwd_kernel": 0.03}
        per, total = bench.algorithmic_bytes(200000, 1080, 1920, 780000, 3)
        roof, valu = bench.build_roofline(kern, per, CFG, total, 0.25e-3, 6300.0)
    finally:
        R._committed_pmc, R.pmc_view_traffic = old, oldt
    assert roof["bound"] == "hbm" and roof["scope"] == "whole_pass" and roof["traffic"] == 123456 + 777
    assert roof["traffic_frames_per_launch"] == 1
    assert roof["algorithmic_bytes_per_view"] == total and roof["peak_spec"] == 8000.0
    assert valu["bound"] == "valu" and valu["kernel"] == "sg_render_bwd_kernel" and valu["valu_wave_instructions_per_launch"] == 8.0e7
    assert abs(valu["frac"] - (8.0e7 / 0.15e-3 / 1e9) / (1024 * 2.4e9 / 2.0 / 1e9)) < 1e-9


def test_other_configuration_or_missing_sidecar_is_not_used(tmp_path):
    meta = {"config": dict(CFG, gaussians=50000), "sources": bench.source_hashes()}
    r = bench._committed_pmc("sg_render_bwd_kernel", CFG, _profiles(tmp_path, meta))
    assert "valu" not in r and "another configuration" in r["stale"]
    (tmp_path / "n").mkdir()
    r = bench._committed_pmc("sg_render_bwd_kernel", CFG, _profiles(tmp_path / "n", None, traffic_meta=None))
    assert "valu" not in r and "no sidecar" in r["stale"] and "traffic" not in r


def test_committed_passes_have_sidecars():
    """Every PMC file the bench may read from profiles/ names its configuration and sources (or is ignored)."""
    pdir = os.path.join(ROOT, "profiles")
    newest = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_SQ.csv"))[-1]
    meta = json.load(open(os.path.join(pdir, newest[:-4] + ".meta.json")))
    assert set(meta["sources"]) >= set(bench.RASTER_SOURCES) and "gaussians" in meta["config"]
    assert "_meta" in json.load(open(os.path.join(pdir, "hbm_traffic.json")))


def test_a_pass_counts_for_a_kernel_only_if_it_is_keyed_to_that_kernels_sources(tmp_path):
    """Round 6: the loss / decode / linear / regulariser / rotation kernels are keyed to their own files.  A sidecar of the
    rasterisation sources alone (rounds 1-5) still serves sg_render_bwd_kernel and says nothing about sg_photo_kernel; a full
    sidecar serves both; a changed sg_loss.hip drops the loss kernel's counts and keeps the composite's."""
    assert set(bench.sources_of("sg_photo_kernel<true>")) == {"sings_amd/csrc/sg_loss.hip", "sings_amd/csrc/sg_common.h"}
    assert bench.sources_of("sg_render_bwd_kernel") == bench.RASTER_SOURCES and bench.sources_of(None, "train") == bench.PMC_SOURCES
    full = bench.source_hashes()
    assert set(full) == set(bench.PMC_SOURCES)
    old = {"config": dict(CFG), "sources": {k: v for k, v in full.items() if k in bench.RASTER_SOURCES}}
    assert bench._meta_status(old, CFG, kernel="sg_render_bwd_kernel") is None
    assert "not keyed to sg_loss.hip" in bench._meta_status(old, CFG, kernel="sg_photo_kernel<true>")
    new = {"config": dict(CFG), "sources": dict(full)}
    assert bench._meta_status(new, CFG, kernel="sg_photo_kernel<true>") is None
    new["sources"]["sings_amd/csrc/sg_loss.hip"] = "0" * 40
    assert "sg_loss.hip" in bench._meta_status(new, CFG, kernel="sg_photo_kernel<true>")
    assert bench._meta_status(new, CFG, kernel="sg_render_bwd_kernel") is None
    assert bench._meta_status(new, dict(CFG, workload="train")) is not None      # (a whole train step: every source counts)


def test_git_blob_hash_is_gits(tmp_path):
    f = tmp_path / "x"
    f.write_bytes(b"hello\n")
    assert bench.git_blob_sha1(str(f)) == "ce013625030ba8dba906f756967f9e9ca394464a"      # git hash-object of "hello\n"


def test_timed_repeats_runs_whole_regions_until_the_minimum(monkeypatch):
    calls = []

    def fake_region(dist, dev, steps, step):
        calls.append(steps)
        return 0.1
    monkeypatch.setattr(bench._distrib, "timed_region", fake_region)
    els = bench.timed_repeats(None, None, 7, None)
    assert calls == [7] * 5 and len(els) == 5                   # 5 x 0.1 s reaches MIN_TIMED_S = 0.5
    calls.clear()
    monkeypatch.setattr(bench._distrib, "timed_region", lambda *a: calls.append(1) or 2.0)
    assert len(bench.timed_repeats(None, None, 3, None)) == 2   # never fewer than two regions
    assert bench._median([3.0, 1.0, 2.0]) == 2.0 and bench._median([1.0, 2.0]) == 1.5


def test_skinned_byte_split_matches_the_survey_formula():
    N, H, W, R, J = 150000, 896, 512, 740000, 52
    per, total = bench.algorithmic_bytes_skinned(N, H, W, R, 0, J)
    base_per, base_total = bench.algorithmic_bytes(N, H, W, R, 0)
    extra_read = N * (12 + 4 * J) - N * 28
    assert per["sg_preprocess_fwd_kernel"] == base_per["sg_preprocess_fwd_kernel"] + extra_read
    assert total == base_total + 2 * extra_read + N * 12 - N * 28
