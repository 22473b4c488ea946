#!/usr/bin/env python3
"""Headline benchmark: rendered views/s, forward+backward, 200k Gaussians @ 1080p (BASELINE.json
configs[2]; SURVEY.md 8(d) scene S(200000,1920,1080,3,seed=3)), 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N fresh rank processes (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; it has not touched the GPU itself), waits, relays rank 0's JSON line and
exits non-zero if any rank failed.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks are already there; either way every rank checks `world == --gpus` and the line carries what the collective layer
saw (`rccl_world`, `dist_backend`).

One "step" = one pass of the hot path over one BATCH of views per rank (default 16 views, `--views-per-step`; every view of a
step carries the same work: cameras a few centimetres apart): the rasterizer forward (preprocess -> tile binning -> alpha
composite) + backward (composite bwd -> per-Gaussian bwd) through the C ABI, inputs resident in HBM, workspaces pre-allocated,
nothing synchronises inside the timed region.  The views go out as launches of `--frames-per-launch` cameras of the same
Gaussians (default 8: ONE dispatch per kernel for the 8 cameras -- the *_frames entry points), the launches are dealt to
`--streams` HIP streams (default 2) and their gradient rows are folded in one pass once they have joined
(sings_amd.dp.GradientPipeline).  `--views-per-step 8 --frames-per-launch 1 --streams 3` is the round-3 schedule (one engine per
view).
`--views-per-step 1 --streams 1` is the reference's one frame per step (gs_trainer.py:207-215); the default run times
that too, after the batched region, and reports it as `train_step_ms_one_view`.  With N > 1 every (rank, view) pair
renders a DIFFERENT camera of the same Gaussians (frame-parallel) and the ranks sum the canonical-Gaussian gradients of
the batch with ONE RCCL all-reduce per step (in chunks, each behind its part of the fold), i.e. the 47 MB all-reduce is paid
once per step of 16 views ("scaling": "weak": the batch per rank is fixed).

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline      - SURVEY.md 8(d): bound "hbm", the WHOLE pass of a view -- algorithmic bytes per view / seconds per view against
                  the float4-copy bandwidth MEASURED in this run (sg_copy_probe over 1 GiB, best of 3; `peak_spec` 8 TB/s rides
                  along), `traffic` = PMC bytes per view; the dominant kernel's own figure (its algorithmic bytes / its live
                  HIP-event duration on the launch stream) as `dominant_kernel_*`
  roofline_valu - the secondary bound: the dominant composite kernel against VALU issue (wave64 VALU instructions per launch
                  from the committed PMC pass x the guide's 2 cycles / (duration x 2.4 GHz x 1024 SIMDs))
  parity        - the three full views the cpu_baseline leg renders with the CPU oracle, COMPARED with the engine's images and
                  gradients for the same cameras (binning bit-exact, RGB L-inf off borderline pixels, worst gradient error)
  cpu_baseline  - PyTorch-CPU "LBS + project" (oracle/lbs_project_torch.py) on all host cores, median of 10 at
                  N = 6 890 / 50 k / 200 k; the scalar C raster oracle (1 core, full views) rides along as an extra key
  allreduce_*   - stand-alone collective time, the part of it the step cannot hide, bytes, per-link bound
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# rooflines / PMC sidecars and the process-group / timing / self-check plumbing live in benchkit/ (no oracle, no workload there);
# the names stay reachable as bench.<name> (tests/test_bench_host.py, tools/)
from benchkit import distrib as _distrib                                     # noqa: E402
from benchkit import roofline as _roofline                                   # noqa: E402
from benchkit.distrib import (COMM_KEYS, FORCE_DIST, LIGHT_TIMED_S, MAX_REPEATS, MIN_TIMED_S, _grad_sha256, _log, _median,   # noqa: E402,F401
                              _ranks_agree, allreduce_probe, densification_stats_check, dist_setup, dp_self_check, one_view_step_by_algorithm,
                              exposed_by_algorithm, make_frame_parallel, spawn_ranks, timed_region, timed_repeats, usable_cores)
from benchkit.baseline import PARITY_BORDER, PARITY_RGB_TOL, _ParityLog, cpu_baseline, cpu_lbs_project, cpu_lbs_project_worker    # noqa: E402,F401
from benchkit.common import _emit, _release, _tile_list_stats                  # noqa: E402,F401
from benchkit.leg_avatar import avatar_parity, leg_avatar                      # noqa: E402,F401
from benchkit.leg_raster import _raster_view, leg_dropin, leg_raster           # noqa: E402,F401
from benchkit.leg_train import leg_train, train_parity                         # noqa: E402,F401
from benchkit.roofline import (CLOCK_HZ, HBM_COPY_GBS, HBM_PEAK_GBS, KERNEL_VARIANTS, PMC_SOURCES, RASTER_SOURCES, SIMDS, VALU_CYCLES_GUIDE,   # noqa: E402,F401
                               VALU_CYCLES_MIX, XGMI_LINK_GBS, _committed_pmc, _meta_status, algorithmic_bytes,
                               algorithmic_bytes_skinned, build_roofline, git_blob_sha1, measure_copy_peak, pmc_view_traffic, train_step_roofline,
                               scaling_model, source_hashes, sources_of)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=200000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--scene-seed", type=int, default=3,
                    help="seed of the synthetic scene S(N,W,H,deg,seed) (SURVEY.md 8(d): cfg2 = 2, cfg3 = 3, cfg5 = 5; what tests/test_gpu_raster.py checks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default single-GPU run only: skip the `secondary` block (the other BASELINE configurations, each a timed "
                         "region of >= 0.3 s in this same process: cfg2 forward, cfg4 avatar, cfg5 + regularisers, the full training "
                         "step at K = 1 and K = 16, the drop-in autograd surface)")
    ap.add_argument("--light", action="store_true",
                    help="a secondary leg's settings: timed regions of >= 0.3 s, one parity view, no LBS + project CPU sweep, no "
                         "collective probe")
    ap.add_argument("--morton", action="store_true", help="avatar / train workloads: store the canonical Gaussians in Morton order")
    ap.add_argument("--forward-only", action="store_true", help="raster workload: time the forward pass only (BASELINE configs[1])")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="train workload: weight-gradient kernels on the backward stream instead of a side stream")
    ap.add_argument("--join-regularisers-early", action="store_true",
                    help="train workload: round-2 schedule (the regularisers' stream forks after the decode and joins before the "
                         "loss sum; default: grids early, k-NN query behind the raster forward, join in the backward pass)")
    ap.add_argument("--graph", action="store_true", help="raster workload: replay the step from a captured HIP graph")
    ap.add_argument("--views-per-step", type=int, default=None,
                    help="raster workload: views each rank renders (gradients summed locally) per step and all-reduce; 1 = the "
                         "reference's one frame per step.  k > 1 amortises the 47 MB all-reduce over k views.  Default: raster 16, avatar 24 (16 is the "
                         "reference's chunk: SinGS.forward_chunk hands the model 16 frames per call)")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="frames (avatar) / cameras (raster) of the same Gaussians rendered by ONE dispatch per kernel (the *_frames "
                         "entry points: K consecutive workspaces, the per-Gaussian backward sums the K frames in registers).  A step "
                         "of --views-per-step views is views / K such batches, dealt to the streams.  1 = one engine per view (round "
                         "3).  Default: 8")
    ap.add_argument("--pipeline", action="store_true",
                    help="avatar workload: instead of one batch of frames per stream, schedule by KIND of kernel on two streams -- the "
                         "composite kernels of all batches on one, everything else beside them on a high-priority one "
                         "(sings_amd.engine.FramePipeline).  Measured and NOT the default: beside a composite kernel the light "
                         "kernels take 3-4x as long (the composites already use 70-77 % of the vector issue slots): 4 970 vs 5 380 "
                         "frames/s (LAB.md)")
    ap.add_argument("--streams", type=int, default=None,
                    help="raster workload: HIP streams the views of one step are spread over (each stream has its own "
                         "workspaces; the per-view gradients are folded on a communication stream)")
    ap.add_argument("--reduce-chunks", type=int, default=4,
                    help="pieces the last view's fold + all-reduce is pipelined in (sings_amd.dp.GradientPipeline)")
    ap.add_argument("--gradient-rows", choices=("streams", "views", "one"), default="streams",
                    help="where the views of a step are summed: one gradient row per stream (default: later views of a stream add to "
                         "their stream's row, the fold reads `streams` rows), one per view (round 2), or one buffer for all")
    ap.add_argument("--one-shot-reduce", action="store_true",
                    help="pre-round-2 schedule for comparison: fold all rows after the last view, then ONE all-reduce")
    ap.add_argument("--regularisers", action="store_true",
                    help="raster workload: also run the geometry-preserving regularisers (exact k-NN Gaussian edge loss + L2Norm, "
                         "value and gradient) once per step on a side stream -- BASELINE configs[4] is "
                         "--gaussians 500000 --width 2048 --height 2048 --regularisers")
    ap.add_argument("--eager", action="store_true", help="train workload: launch the ~600 kernels of a step from Python instead of "
                    "replaying the step from a captured HIP graph (the default; host-speed independent)")
    ap.add_argument("--grad-hash", action="store_true",
                    help="raster / avatar workloads: after the timed region run step 0 once more and report the sha256 of the reduced "
                         "gradient buffer (`grad_sha256`; the backward is bitwise reproducible)")
    ap.add_argument("--workload", choices=("raster", "avatar", "train"), default="raster",
                    help="raster = BASELINE configs[2] (the metric's config, default); avatar = configs[3]: ~150k canonical "
                         "Gaussians, J=52, AMASS frames, 512x896, LBS-fused kernels (reported as an extra workload)")
    a = ap.parse_args()
    # per-workload defaults (measured on one MI355X: tools/r04_combos.sh, r04_raster_combos.sh; LAB.md): 16 views per step as two
    # launches of 8 frames / cameras on 2 streams.  raster: 4 088-4 131 views/s (8 views, one camera per launch, 3 streams -- the round-3
    # schedule, --views-per-step 8 --frames-per-launch 1 --streams 3 --: 3 887; one launch of 8: 4 085); avatar: 5 820-5 845 frames/s
    # (one launch of 8: 4 940; 24 frames on 3 streams: 5 930-6 005)
    # avatar: 24 frames = three launches of 8 on three streams -- 5 930-6 010 against 5 770-5 940 frames/s for 16 / 2: the latency-bound
    # per-Gaussian kernels of an avatar end in partly filled rounds (2 344 waves on 2 048 slots) which a third stream fills; the
    # raster step gains nothing from a third stream (its composites saturate the vector issue on their own)
    if a.views_per_step is None:
        a.views_per_step = 24 if a.workload == "avatar" else 1 if a.workload == "train" else 16     # (train: the reference's one frame per step)
    if a.streams is None:
        a.streams = 3 if a.workload == "avatar" else 2
    return a


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--_cpu-worker":
        return cpu_lbs_project_worker(sys.argv[2:])
    a = parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a, os.path.abspath(__file__))
    t_wall = time.perf_counter()
    ctx = dist_setup(a)
    out = LEGS[a.workload](a, ctx)                               # rank 0: the line (a dict); other ranks: None
    if out is not None and wants_secondary(a, ctx):
        out["secondary"] = secondary_legs(a, ctx, out)
    if ctx[3] is not None:
        ctx[3].destroy_process_group()
    if out is not None:
        _log("done")
        out["wall_s"] = time.perf_counter() - t_wall
        _emit(out)
        # a multi-GPU run is short and decisive or it is an error: one rank per device over RCCL (dist_setup refuses anything else on a
        # box with enough GPUs), the reduced bytes equal on all ranks and equal to the re-rendered sum, the whole run inside MULTI_GPU_WALL_S
        if ctx[1] > 1 and not ctx[4].get("ranks_per_device", 1) > 1:
            if ctx[4].get("rccl_world") != ctx[1] or out.get("ranks_agree") is False or out["wall_s"] > MULTI_GPU_WALL_S:
                raise SystemExit(f"bench.py: the {ctx[1]}-GPU run is not decisive: rccl_world={ctx[4].get('rccl_world')}, "
                                 f"ranks_agree={out.get('ranks_agree')}, wall {out['wall_s']:.0f} s (limit {MULTI_GPU_WALL_S} s)")


MULTI_GPU_WALL_S = 120               # --gpus N > 1: the whole run (all ranks up, warm-up, timed regions, checks) must end inside this


def wants_secondary(a, ctx):
    """The default single-GPU headline run (what the driver launches) also measures every other BASELINE configuration."""
    return (ctx[1] == 1 and not a.no_secondary and not a.no_cpu_baseline and not a.light and a.workload == "raster" and not a.forward_only
            and not a.graph and (a.gaussians, a.width, a.height, a.sh_degree) == (200000, 1920, 1080, 3) and not a.regularisers)


LEGS = {"raster": leg_raster, "avatar": leg_avatar, "train": leg_train}


def _compact(j):
    """What a secondary leg contributes to the headline line."""
    keep = {k: j.get(k) for k in ("metric", "value", "unit", "ms_per_step", "ms_per_view", "train_step_ms_one_view", "timed_region_s",
                                  "repeats", "parity", "losses", "schedule_note") if k in j}
    cfg = j.get("config", {})
    keep["config"] = {k: cfg[k] for k in ("workload", "gaussians", "width", "height", "num_rendered", "max_num_rendered", "tile_list_mean",
                                          "tile_list_max", "views_per_step", "frames_per_launch", "streams", "frames_per_step",
                                          "regularisers", "forward_only", "hip_graph") if k in cfg}
    r = j.get("roofline")
    if r:
        keep["roofline"] = {k: r.get(k) for k in ("bound", "scope", "achieved", "peak", "unit", "frac", "frac_of_spec", "algorithmic_bytes_per_view",
                                                  "traffic", "dominant_kernel", "dominant_kernel_ms", "dominant_kernel_frac") if k in r}
        for sub in ("mfma", "hbm"):                                 # (the train step's two roofs)
            if isinstance(r.get(sub), dict):
                keep["roofline"][sub] = {k: v for k, v in r[sub].items() if k != "note"}
    if "raster_oracle_1core" in (j.get("cpu_baseline") or {}):
        keep["cpu_oracle_views_per_s_1core"] = j["cpu_baseline"]["raster_oracle_1core"]["value"]
    elif (j.get("cpu_baseline") or {}).get("unit") == "views/s":
        keep["cpu_oracle_views_per_s_1core"] = j["cpu_baseline"]["value"]
    return keep


def secondary_legs(a, ctx, head):
    """The other BASELINE configurations, in this process, after the headline measurement (`--no-secondary` skips them): each a
    timed region of >= 0.3 s with its own algorithmic bytes against the copy rate measured in THIS run, R / tile-list statistics
    and a parity block from one full-size oracle view."""
    import copy
    t0 = time.perf_counter()
    base = copy.copy(a)
    base.light, base.no_secondary, base.copy_gbs = True, True, head.get("hbm_copy_GBs_measured")
    base.grad_hash = False

    def variant(**kw):
        v = copy.copy(base)
        for k, val in kw.items():
            setattr(v, k, val)
        return v
    plan = [
        ("cfg2_forward", "raster", variant(gaussians=50000, width=512, height=512, sh_degree=0, scene_seed=2, forward_only=True, steps=max(a.steps, 50))),
        ("cfg4_avatar", "avatar", variant(workload="avatar", views_per_step=24, streams=3, frames_per_launch=None, steps=max(a.steps // 2, 10))),
        ("cfg5_regularisers", "raster", variant(gaussians=500000, width=2048, height=2048, scene_seed=5, regularisers=True, steps=max(a.steps // 4, 5))),
        ("train_step_K1", "train", variant(workload="train", views_per_step=1, steps=max(a.steps, 20), warmup=5)),
        ("train_step_K16", "train", variant(workload="train", views_per_step=16, steps=max(a.steps // 2, 10), warmup=3, no_cpu_baseline=True)),
    ]
    sec = {}
    for name, leg, args in plan:
        _release()
        _log(f"secondary leg {name}")
        try:
            sec[name] = _compact(LEGS[leg](args, ctx))
        except Exception as e:                                       # a failing extra must not cost the headline line
            sec[name] = {"error": f"{type(e).__name__}: {e}"}
    _release()
    _log("secondary leg dropin_autograd")
    try:
        sec["dropin_autograd"] = leg_dropin(base, ctx)
        sec["dropin_autograd_ms_per_view"] = sec["dropin_autograd"]["ms_per_view"]
    except Exception as e:
        sec["dropin_autograd"] = {"error": f"{type(e).__name__}: {e}"}
    if isinstance(sec.get("train_step_K16"), dict) and sec["train_step_K16"].get("parity") is None:
        # (same model, same kernels and frame 0 as train_step_K1, whose parity block stands for both; the chunk itself:)
        sec["train_step_K16"]["parity"] = {"see": "train_step_K1.parity (same decode, frame 0 of the chunk)",
                                           "chunk": "tests/test_gpu_train_step.py::test_a_step_over_a_chunk_of_frames_equals_the_"
                                                    "sum_of_one_frame_steps (loss and every gradient of a K-frame step = the sum of K one-frame steps)"}
    if "train_step_ms_one_view" in sec.get("cfg4_avatar", {}):
        sec["cfg4_avatar"]["frames_per_s_one_frame_per_step"] = 1e3 / sec["cfg4_avatar"]["train_step_ms_one_view"]
    _release()
    sec["wall_s"] = time.perf_counter() - t0
    return sec


if __name__ == "__main__":
    main()
