"""bench.py's command line contract, run as the driver runs it (child processes, last stdout line = JSON)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gaussians", "20000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1"]


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e,
                       timeout=timeout, cwd=ROOT)
    return p


def _line(p):
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_single_gpu_line_carries_the_contract():
    j = _line(_bench("--gpus", "1", *SMALL))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_valu", "cpu_baseline", "parity",
                "train_step_ms_one_view"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["unit"] == "views/s" and j["dtype"] == "f32" and j["vs_baseline"] is None
    # the HBM roofline of the whole pass against the float4-copy bandwidth measured in this run
    r = j["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] == "hbm" and r["scope"] == "whole_pass" and r["unit"] == "GB/s" and "measured in this run" in r["peak_source"]
    assert 3000.0 < r["peak"] < 8000.0 and r["peak"] == j["hbm_copy_GBs_measured"] and r["peak_spec"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_view"] / (r["ms_per_view"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["dominant_kernel"].startswith("sg_") and 0 < r["dominant_kernel_frac"] < 1
    # the oracle's three full views are COMPARED with the engine's (same cameras): the benchmark proves its own parity
    par = j["parity"]
    assert par["views"] == 3 and par["ok"] and par["binning_exact"] and par["rgb_linf"] <= 1e-5
    assert par["grad_violations"] == 0 and par["borderline_px_beyond_flip_bound"] == 0 and not par["errors"]
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] in (cb["usable_cores"], 16) and cb["usable_cores"] <= os.cpu_count()
    assert set(cb["median_ms_by_points"]) == {"6890", "50000", "200000", "20000"}         # SURVEY 8(d)'s three sizes + the run's own N
    assert cb["raster_oracle_1core"]["cores"] == 1


@pytest.mark.gpu
def test_timed_region_is_repeated_and_the_schema_is_one():
    """--steps 3 of a small scene is a few milliseconds: the region is repeated (whole regions of 3 steps) until it adds up to
    0.5 s, the line reports the median; the collective keys exist (null) without a collective."""
    j = _line(_bench("--gpus", "1", "--no-cpu-baseline", *SMALL))
    assert j["repeats"] >= 2 and j["timed_region_s"] >= 0.5 and j["steps"] == 3
    assert j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    for k in ("allreduce_ms", "allreduce_bytes", "allreduce_algorithm", "allreduce_per_link_bound_ms", "allreduce_exposed_ms"):
        assert k in j and j[k] is None
    assert j["rccl_world"] is None and j["dist_backend"] is None
    # the small scene has no committed PMC pass: the composite kernel's VALU roofline is omitted, with the reason
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["traffic"] is None
    if j["roofline"]["dominant_kernel"] in ("sg_render_bwd_kernel", "sg_render_fwd_kernel"):
        assert j["roofline_valu"]["frac"] is None and "PMC" in j["roofline_valu"]["note"]


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["all_reduce", "rs_ag"])
def test_rccl_branches_run_with_one_rank(algo):
    """SINGS_BENCH_FORCE_DIST=1: init_process_group("nccl", device_id=...), FrameParallel's asynchronous all-reduce / in-place
    reduce-scatter + all-gather on device views, GradientPipeline's chunked fold + collective with RCCL work handles and the
    device-side MAX of the timed region all EXECUTE, on a one-rank communicator -- and leave the gradients bit-identical to
    the run without a process group (a one-rank sum is the identity; the backward is bitwise reproducible)."""
    ref = _line(_bench("--gpus", "1", "--no-cpu-baseline", "--grad-hash", *SMALL))
    j = _line(_bench("--gpus", "1", "--no-cpu-baseline", "--grad-hash", *SMALL,
                     env={"SINGS_BENCH_FORCE_DIST": "1", "SINGS_DP_ALGO": algo}))
    assert j["dist_backend"] == "nccl" and j["rccl_world"] == 1 and j["dist_world"] == 1
    assert j["allreduce_algorithm"] == algo and j["allreduce_ms"] > 0 and j["allreduce_exposed_ms"] > 0
    assert j["allreduce_bytes"] == 20000 * 59 * 4
    assert "4 chunk" in j["config"]["reduction"]
    assert j["grad_sha256"] == ref["grad_sha256"]
    assert j["train_step_ms_one_view"] > 0 and j["value"] > 0


@pytest.mark.gpu
def test_rccl_branches_run_with_one_rank_avatar_and_train():
    """The same for the other two workloads: the avatar step (fused LBS path, 3 + 3 + 1 + 3 * 16 floats per Gaussian all-reduced) and the
    complete training step (parameter gradients written straight into ONE flat buffer -- sings_amd.decode.set_gradient_arena --
    and all-reduced in place)."""
    small = ["--gaussians", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = {"SINGS_BENCH_FORCE_DIST": "1"}
    ref = _line(_bench("--workload", "avatar", "--grad-hash", *small))
    j = _line(_bench("--workload", "avatar", "--grad-hash", *small, env=env))
    assert j["dist_backend"] == "nccl" and j["rccl_world"] == 1 and j["allreduce_ms"] > 0
    # SH gradients coefficient-major: of the 55 gradient floats per Gaussian the 10 that carry gradient at sh_degree 0 are all-reduced
    assert j["allreduce_bytes"] == 20000 * 10 * 4 and j["grad_sha256"] == ref["grad_sha256"]
    assert j["gradient_floats_per_gaussian"] == dict(j["gradient_floats_per_gaussian"], buffer=55, carrying_gradient=10)
    m = j["scaling_model"]
    assert m["collective_bytes"] == 20000 * 10 * 4 and "MODEL" in m["label"] and 0 < m["predicted_scale_8"]["one_view_per_step"]["rs_ag"] <= 8
    assert j["repeats"] >= 2 and j["roofline"]["dominant_kernel"].startswith("sg_") and j["train_step_ms_one_view"] > 0
    ref = _line(_bench("--workload", "train", *small))
    assert ref["allreduce_ms"] is None and ref["rccl_world"] is None
    j = _line(_bench("--workload", "train", *small, env=env))
    assert j["dist_backend"] == "nccl" and j["rccl_world"] == 1 and j["allreduce_ms"] > 0
    assert j["allreduce_bytes"] == 4 * j["config"]["trainable_parameters"]
    # the tri-plane planes and the decoder weights (> 99 % of the parameters) were written in place, the rest copied
    assert j["config"]["gradient_bytes_written_in_place"] >= 0.95 * j["allreduce_bytes"]
    for k, v in ref["losses"].items():                               # (the tri-plane backward uses float atomics: not bitwise)
        assert abs(j["losses"][k] - v) <= 1e-4 * max(1.0, abs(v)), k


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_and_the_collective_layer_sees_them():
    """`python bench.py --gpus 2` must really run two ranks (round 1 ignored the flag).  A one-GPU box is oversubscribed:
    both ranks share the device and the collectives are host-staged gloo (RCCL refuses two ranks on one device); on a
    box with >= 2 GPUs the same command runs over RCCL and `rccl_world` is 2."""
    import torch
    j = _line(_bench("--gpus", "2", "--no-cpu-baseline", *SMALL))
    assert j["n_gpus"] == 2 and j["dist_world"] == 2
    assert j["config"]["parallelism"] == "dp2"
    if torch.cuda.device_count() >= 2:
        assert j["rccl_world"] == 2 and j["dist_backend"] == "nccl" and j["ranks_per_device"] == 1
    else:
        assert j["rccl_world"] is None and j["dist_backend"] == "gloo" and j["ranks_per_device"] == 2
    assert j["allreduce_bytes"] == 20000 * 59 * 4 and j["allreduce_ms"] > 0 and j["allreduce_exposed_ms"] > 0
    assert j["value"] > 0
    # what makes a real N > 1 run decisive, exercised here at world 2:
    #  (a) every rank holds the same reduced bytes; rank 0 re-rendered all 2 x 16 views itself and found the same sum
    assert j["ranks_agree"] is True
    assert j["dp_parity"]["ok"] and j["dp_parity"]["views"] == 2 * j["config"]["views_per_step"] and j["dp_parity"]["violations"] == 0
    #  (b) both collectives timed stand-alone and exposed in the SAME run, the faster one ran the step
    by = j["allreduce_ms_by_algorithm"]
    assert set(by) == {"all_reduce", "rs_ag"} and min(by.values()) > 0
    assert j["allreduce_algorithm"] == min(by, key=by.get) and "measured" in j["allreduce_algorithm_chosen_by"]
    assert set(j["allreduce_exposed_ms_by_algorithm"]) == {"all_reduce", "rs_ag"}
    #  (c) one device per rank, or the line says that the box is oversubscribed
    #  (d) round 6: the reference's one view per step with EACH collective next to the batched step, and the whole run short
    #      (a real multi-GPU run that takes longer than bench.MULTI_GPU_WALL_S exits non-zero)
    ov = j["one_view_per_step_by_algorithm"]
    assert set(ov) == {"all_reduce", "rs_ag"} and all(v["ms_per_step"] > 0 and v["views_per_s"] > 0 for v in ov.values())
    assert 0 < j["wall_s"] < 120
    assert len(j["rank_devices"]) == 2
    assert (len(set(j["rank_devices"])) == 2) == (torch.cuda.device_count() >= 2)


@pytest.mark.gpu
def test_gpus_2_avatar_reduces_the_densification_statistics_and_the_ranks_agree():
    """The avatar workload at world 2: gradient sum checked as above, and the densification statistics (sum of |viewspace gradient|,
    visibility count, max radius: sings_hybrid.py:1013-1015, gs_trainer.py:486-492) reduced and equal on both ranks."""
    j = _line(_bench("--gpus", "2", "--workload", "avatar", "--gaussians", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                     "--views-per-step", "8", "--frames-per-launch", "4"))
    assert j["n_gpus"] == 2 and j["dist_world"] == 2 and j["ranks_agree"] is True
    assert j["dp_parity"]["ok"] and j["dp_parity"]["views"] == 16
    d = j["densification_stats"]
    assert d["reduced"] and d["ranks_agree"] and d["visible_sum"] > 0 and d["max_radius"] > 0
    assert j["allreduce_bytes"] == 20000 * 10 * 4


@pytest.mark.gpu
def test_default_line_carries_every_baseline_configuration():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command): cfg3 stays `value`; `secondary` holds the other
    BASELINE configurations, each from a timed region of >= 0.3 s in the same process, with its own roofline against the copy rate
    of the run and a parity block from a full-size oracle view."""
    j = _line(_bench("--gpus", "1", "--steps", "20", "--warmup", "5", timeout=1500))
    assert j["config"]["gaussians"] == 200000 and j["parity"]["ok"] and j["parity"]["views"] == 3
    assert j["train_step_ms_one_view"] > j["raster_fwd_bwd_ms_one_view"] > 0            # (the loss is inside the one-view train step)
    sec = j["secondary"]
    for k in ("cfg2_forward", "cfg4_avatar", "cfg5_regularisers", "train_step_K1", "train_step_K16", "dropin_autograd"):
        assert k in sec and "error" not in sec[k], (k, sec.get(k))
    for k in ("cfg2_forward", "cfg4_avatar", "cfg5_regularisers"):
        e = sec[k]
        assert e["value"] > 0 and e["timed_region_s"] >= 0.3 and e["repeats"] >= 2
        r = e["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == j["hbm_copy_GBs_measured"] and 0 < r["frac"] < 1
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["algorithmic_bytes_per_view"] > 0
        assert e["parity"]["ok"], (k, e["parity"])
    assert sec["cfg2_forward"]["config"]["forward_only"] and sec["cfg2_forward"]["config"]["gaussians"] == 50000
    # the SURVEY 8(d) scenes, the ones tests/test_gpu_raster.py checks: cfg2 = S(..., seed 2), cfg3 = seed 3, cfg5 = seed 5
    assert "seed=3)" in j["config"]["workload"] and "seed=2)" in sec["cfg2_forward"]["config"]["workload"]
    assert "seed=5)" in sec["cfg5_regularisers"]["config"]["workload"]
    assert sec["cfg5_regularisers"]["config"]["regularisers"] and sec["cfg5_regularisers"]["config"]["gaussians"] == 500000
    av = sec["cfg4_avatar"]
    assert av["config"]["gaussians"] == 150000 and av["train_step_ms_one_view"] > 0 and av["frames_per_s_one_frame_per_step"] > 0
    assert av["parity"]["binning_exact"] and av["parity"]["gradient_arrays_compared"] == 7
    for k, K in (("train_step_K1", 1), ("train_step_K16", 16)):
        e = sec[k]
        assert e["config"]["frames_per_step"] == K and e["value"] > 0 and e["timed_region_s"] >= 0.3 and "trajectory" in e["schedule_note"]
    assert sec["train_step_K1"]["parity"]["ok"], sec["train_step_K1"]["parity"]
    for k in ("train_step_K1", "train_step_K16"):                   # round 6: the step has a stated bound (decoder FLOPs on the fp32 matrix cores, bytes)
        r = sec[k]["roofline"]
        assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["mfma"]["flops_per_step"] > 5e10 and r["hbm"]["algorithmic_bytes_per_step"] > 3e9
        assert abs(r["frac"] - max(r["mfma"]["frac"], r["hbm"]["frac"])) < 1e-9
    assert sec["dropin_autograd_ms_per_view"] == sec["dropin_autograd"]["ms_per_view"] > 0
    assert sec["wall_s"] < 240


@pytest.mark.gpu
def test_world_size_mismatch_is_an_error():
    p = _bench("--gpus", "2", *SMALL, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stdout + p.stderr)


def test_gpus_flag_spawns_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert p.returncode != 0
    assert "rank exit codes" in p.stderr and "no CPU path" in p.stderr
