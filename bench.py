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


def _release():
    """Between legs of one process: drop what the finished leg allocated (its locals are gone) and reset the library's global modes."""
    import gc
    import torch
    from sings_amd import rasterizer as _rz
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    _rz.set_overflow_check("sync")
    _rz.reset_overflow_state()


MULTI_GPU_WALL_S = 120               # --gpus N > 1: the whole run (all ranks up, warm-up, timed regions, checks) must end inside this


def wants_secondary(a, ctx):
    """The default single-GPU headline run (what the driver launches) also measures every other BASELINE configuration."""
    return (ctx[1] == 1 and not a.no_secondary and not a.no_cpu_baseline and not a.light and a.workload == "raster" and not a.forward_only
            and not a.graph and (a.gaussians, a.width, a.height, a.sh_degree) == (200000, 1920, 1080, 3) and not a.regularisers)


def leg_raster(a, ctx):
    """BASELINE configs[2] (default), configs[1] (--forward-only, 50 k @ 512^2, degree 0), the raster part of configs[4].
    -> the JSON line as a dict on rank 0, None on the other ranks."""
    import numpy as np
    import torch
    rank, world, dev, dist, dinfo = ctx

    from sings_amd import _lib
    from sings_amd.engine import RasterEngine, ViewBatch
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene

    N, W, H, deg = a.gaussians, a.width, a.height, a.sh_degree
    seed = int(getattr(a, "scene_seed", 3))
    s = synthetic_scene(N, W, H, deg, seed)
    # frame-parallel: every (rank, view-of-the-step) pair looks at the same Gaussians from its own camera, shifted along x
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    bg_t = t(s["bg"])

    def camera(index):
        # the views of a step: the scene's camera displaced by a few centimetres -- distinct cameras that all carry the SAME work
        # (R within 0.1 % of camera 0's).  Rounds 1-3 shifted by 0.05 x index: from index ~4 on the scene slides out of the frustum
        # (camera 7: R - 3.7 %, camera 15: - 13 %, camera 63: - 64 %), i.e. a larger batch rendered LIGHTER views (LAB.md 4.7)
        view = s["viewmatrix"].copy()
        view[3, 0] += 0.012 * (index % 8)
        view[3, 1] += 0.012 * ((index // 8) % 8)
        view[3, 0] += 0.0015 * (index // 64)                      # beyond 64 cameras (ranks >= 4 at 16 views): still distinct
        proj = (view @ P_T).astype(np.float32)
        campos = np.linalg.inv(view)[3, :3].astype(np.float32)
        return view, proj, campos, GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=bg_t, scale_modifier=1.0,
            viewmatrix=t(view), projmatrix=t(proj), sh_degree=deg, campos=t(campos), prefiltered=False, debug=False)

    k_views = 1 if a.graph else max(1, a.views_per_step)       # a captured graph replays one view on one stream
    view, proj, campos, rs = camera(rank * k_views)
    means3D, shs, opac, scales, rots = t(s["means3D"]), t(s["shs"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])
    dL = t(s["dL_dimage"])

    # sizing pass (untimed): find R, then fix the pair capacity for the whole run
    _log(f"scene ready (world {world}); sizing pass")
    eng = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=8 * N + 65536)
    eng.set_camera(rs)
    R = eng.forward(means3D, shs, opac, scales, rots, sync_num_rendered=True)
    if R > eng.cap:
        raise SystemExit(f"pair capacity too small: R={R}")
    tile_mean, tile_max = _tile_list_stats(eng, W, H)
    del eng
    torch.cuda.empty_cache()

    fp, algo_info = make_frame_parallel(ctx, N * (3 + 3 + 4 + 1 + 3 * shs.shape[1]))

    # K cameras per launch (round 4, sings_amd.engine.RasterFramesEngine): the step's k_views views go out as k_views / K batches
    Kf = 1 if a.graph else max(1, min(a.frames_per_launch if a.frames_per_launch is not None else 8, k_views, _lib.MAX_FRAMES))
    while k_views % Kf:
        Kf -= 1
    n_batches = k_views // Kf
    n_streams = max(1, min(a.streams, n_batches))
    per_view = N * (3 + 3 + 4 + 1 + 3 * shs.shape[1])
    # the k views of a step: each has its own engine (= workspaces, so that views in flight at the same time on different
    # streams share no state) writing its gradients into its own row of `grads`; ViewBatch deals them to the streams and
    # folds the rows (+ all-reduce) on a communication stream
    # round 3: one gradient row per STREAM -- the first view of a stream writes it, the later ones add to it (accumulate mode of
    # the per-Gaussian backward), so the fold after the join reads `streams` rows, not `views` rows.  --gradient-rows views: the
    # round-2 scheme; one: a single buffer, the views' last kernels ordered across the streams by events.
    rows = {"streams": n_streams, "views": n_batches, "one": 1}[a.gradient_rows]
    grads = ViewBatch.gradient_rows(rows, per_view, dev)
    engs = []
    short = tile_max * 1.5 <= 1024
    for v in range(n_batches):
        # the sizing pass knows the longest tile list (135 at cfg3): with 1.5x margin for the other cameras of the batch no
        # list can need the long-list sort kernels (lists <= 1024 are sorted by the compositing workgroups; checked on the device, a
        # violation surfaces in num_rendered() below)
        if Kf == 1:
            e = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=int(R * 1.1) + 4096, grad_flat=grads[v % rows])
            e.set_camera(camera(rank * k_views + v)[3], short_lists=short)
        else:
            from sings_amd.engine import RasterFramesEngine
            e = RasterFramesEngine(N, W, H, shs.shape[1], Kf, dev, capacity_pairs=int(R * 1.1) + 4096, grad_flat=grads[v % rows])
            cams = [camera(rank * k_views + v * Kf + f) for f in range(Kf)]
            e.set_camera(cams[0][3]._replace(viewmatrix=t(np.stack([c_[0] for c_ in cams])), projmatrix=t(np.stack([c_[1] for c_ in cams])),
                                             campos=t(np.stack([c_[2] for c_ in cams]))), short_lists=short)
        engs.append(e)
    if Kf == 1:
        eng = engs[0]
    else:                                                        # the one-view-per-step leg and the parity views: a plain engine
        eng = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=int(R * 1.1) + 4096)
        eng.set_camera(camera(rank * k_views)[3], short_lists=short)
    dL_k = dL if Kf == 1 else dL[None].expand(Kf, -1, -1, -1).contiguous()
    batch = ViewBatch(engs, grads, n_streams, frame_parallel=fp, chunks=a.reduce_chunks)

    graph = eng.capture(means3D, shs, opac, scales, rots, None if a.forward_only else dL) if a.graph else None

    def one_view(v, e):
        e.forward(means3D, shs, opac, scales, rots)
        if not a.forward_only:
            e.backward(means3D, shs, opac, scales, rots, dL if getattr(e, "K", 1) == 1 else dL_k)

    reg = None
    if a.regularisers:
        # per optimisation step, not per view: Gaussian positions / scales / opacities are the same for all views
        from sings_amd.regularizers import GaussiansEdgeLoss, L2Norm
        reg_mods = (GaussiansEdgeLoss(), L2Norm())
        reg_sc = scales.clone().requires_grad_(True); reg_off = (0.002 * torch.randn_like(means3D)).requires_grad_(True)
        reg_side = torch.cuda.Stream(dev)

        def reg():
            reg_sc.grad = None; reg_off.grad = None
            loss = reg_mods[0]({"xyz_canon": means3D, "scales": reg_sc}) + reg_mods[1]({"xyz_offsets": reg_off, "scales": reg_sc,
                                                                                        "opacity": opac})
            loss.backward()

    def step(_i=0):
        if reg is not None:
            cur = torch.cuda.current_stream(dev)
            reg_side.wait_stream(cur)
            with torch.cuda.stream(reg_side):
                reg()
        if graph is not None:
            graph.replay()
            if fp is not None:
                fp.all_reduce_grads(eng.grad_flat)
        elif a.one_shot_reduce:
            batch.run_unreduced(one_view)
            batch.pipe.one_shot()
        else:
            batch.run(one_view)
        if reg is not None:
            cur.wait_stream(reg_side)

    _log(f"R = {R}; warm-up ({a.warmup} steps of {k_views} views on {n_streams} streams)")
    for _ in range(a.warmup):
        step()
    if os.environ.get("SINGS_BENCH_HOSTTIME"):                    # how long does the host take to SUBMIT a step? (GPU idle at the start)
        import time as _t
        for _ in range(3):
            torch.cuda.synchronize(dev); t0 = _t.perf_counter(); step(); t1 = _t.perf_counter(); torch.cuda.synchronize(dev)
            t2 = _t.perf_counter()
            _log(f"host submission {1e3 * (t1 - t0):.3f} ms, step complete after {1e3 * (t2 - t0):.3f} ms")
    _log(f"timed region ({a.steps} steps, repeated until {MIN_TIMED_S} s)")
    els = timed_repeats(dist, dev, a.steps, step, min_s=LIGHT_TIMED_S if a.light else None)
    el = _median(els)
    _log(f"{el / a.steps * 1e3:.3f} ms per step (median of {len(els)} regions); one view per step")
    assert all(0 <= r_ <= e.cap for e in engs for r_ in (e.num_rendered() if getattr(e, "K", 1) > 1 else [e.num_rendered()])), \
        "pair capacity / short-list hint violated"
    ms_per_step = el / a.steps * 1e3
    views_s = world * a.steps * k_views / el
    grad_hash = None
    if a.grad_hash:
        step()
        torch.cuda.synchronize()
        grad_hash = _grad_sha256(eng.grad_flat if graph is not None else batch.acc)

    # the reference's unit of work, one frame per optimisation step (gs_trainer.py:207-215): view 0 alone on the current
    # stream, (+ the all-reduce of its gradients with several ranks), same number of views as the batched region
    def step_one_view(_i=0):
        one_view(0, eng)
        if fp is not None:
            fp.all_reduce_grads(eng.grad_flat)
    n_one = max(20, min(a.steps * k_views, 2000))
    eng.throughput = False                                       # one view in flight from here on: the library may spend work on latency
    for _ in range(10):
        step_one_view()
    el_one = _median(timed_repeats(dist, dev, n_one, step_one_view, min_s=0.25))
    one_by_algo = None
    if fp is not None and world > 1:
        _log("one view per step with each collective")
        one_by_algo = one_view_step_by_algorithm(ctx, fp, eng.grad_flat, lambda: one_view(0, eng), max(20, n_one // 4), timed_repeats, _median)
    # SURVEY.md 8(d) "Timing": train-step ms = forward + L1(-SSIM)-to-random-target loss + backward (+ the all-reduce): the same
    # one-view step with the photometric loss of the reference (clamp, 0.8 L1 + 0.2 SSIM: loss.py:55-69) computed from the rendered
    # image and ITS gradient fed to the backward, instead of a fixed dL/dimage
    el_loss = None
    if not a.forward_only:
        from sings_amd.photo_loss import PhotoLossEngine
        loss1 = PhotoLossEngine(W, H, dev, l1_w=0.8, ssim_w=0.2)
        torch.manual_seed(0)
        gt_rgb = torch.rand((3, H, W), device=dev); ones = torch.ones((H, W), device=dev)

        def step_one_view_loss(_i=0):
            eng.forward(means3D, shs, opac, scales, rots)
            eng.backward(means3D, shs, opac, scales, rots, loss1(eng.color, gt_rgb, ones, bg_t))
            if fp is not None:
                fp.all_reduce_grads(eng.grad_flat)
        for _ in range(10):
            step_one_view_loss()
        el_loss = _median(timed_repeats(dist, dev, n_one, step_one_view_loss, min_s=0.25))

    # collective: stand-alone time and the part of it the batched step cannot hide
    _log("collective probe / per-kernel event pass")
    comm = allreduce_probe(fp, batch.acc)
    if comm is not None and graph is None and not a.one_shot_reduce:
        batch.pipe.enable_timing(True)
        ex = []
        for _ in range(10):
            torch.cuda.synchronize()
            dist.barrier()
            step()
            torch.cuda.synchronize()
            ex.append(batch.pipe.exposed_ms())
        batch.pipe.enable_timing(False)
        comm["allreduce_exposed_ms"] = sorted(ex)[len(ex) // 2]
        comm["allreduce_hidden_note"] = (f"the fold of the k rows is pipelined with the collective in {len(batch.pipe.bounds)} "
                                         "chunks; exposed = last view's gradients ready -> reduced sum ready (median of 10 "
                                         "synchronised steps; includes the fold)")

    # per-kernel durations: HIP events around every launch, on the launch stream (separate pass, one view at a time)
    lib = _lib.load()
    lib.sg_profile_enable(1)
    for _ in range(max(20, min(a.steps, 100))):
        one_view(0, eng)
    ms = (C.c_double * _lib.NUM_KERNELS)()
    cnt = (C.c_int64 * _lib.NUM_KERNELS)()
    _lib.check(lib.sg_profile_collect(ms, cnt, _lib.NUM_KERNELS), "profile")
    lib.sg_profile_enable(0)
    kern = {lib.sg_kernel_name(k).decode(): (ms[k] / max(cnt[k], 1)) for k in range(_lib.NUM_KERNELS)}

    dp_check = None
    if dist is not None and not a.graph:
        dp_check = dp_self_check(ctx, lambda: (step(), batch.acc)[1],
                                 lambda v: _raster_view(eng, camera, v, (means3D, shs, opac, scales, rots), s["dL_dimage"], t, host=False)["flat"],
                                 world * k_views)
        dp_check.update(algo_info)
        if comm is not None and not a.one_shot_reduce and world > 1:
            dp_check["allreduce_exposed_ms_by_algorithm"] = exposed_by_algorithm(ctx, batch.pipe, step)
    if rank != 0:
        return None

    per, total_bytes = algorithmic_bytes(N, H, W, R, deg)
    if a.forward_only:                                          # B_f = N (in + 4 + 2 rec) + HW 12 + R 16
        total_bytes = N * (44 + 12 * (deg + 1) ** 2 + 4 + 2 * 75) + H * W * 12 + R * 16
    _log("float4-copy probe (the roofline's denominator)")
    copy_gbs = a.copy_gbs if getattr(a, "copy_gbs", None) else measure_copy_peak(dev)
    roofline, roofline_valu = build_roofline(kern, per, {"workload": "raster", "gaussians": N, "width": W, "height": H, "sh_degree": deg},
                                             total_bytes, world / views_s, copy_gbs, frames=Kf)
    out = {
        "metric": "rendered views/sec fwd+bwd, 200k Gaussians @1080p" if not a.forward_only and (N, W, H) == (200000, 1920, 1080)
                  else f"rendered views/sec {'forward only' if a.forward_only else 'fwd+bwd'}, {N} Gaussians @{W}x{H}",
        "value": views_s, "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "ms_per_view": ms_per_step / k_views,
        "train_step_ms_one_view": (el_loss if el_loss is not None else el_one) / n_one * 1e3,
        "train_step_ms_one_view_note": "one view per step: forward + clamp / 0.8 L1 + 0.2 SSIM loss against a random target + backward "
                                       "(+ the all-reduce with several ranks): SURVEY.md 8(d) Timing" if el_loss is not None else
                                       "forward only (no loss, no backward)",
        "raster_fwd_bwd_ms_one_view": el_one / n_one * 1e3, "views_per_s_one_view_per_step": world * n_one / el_one,
        "timed_region_s": sum(els), "repeats": len(els), "ms_per_step_min": min(els) / a.steps * 1e3,
        "ms_per_step_max": max(els) / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"S({N},{W},{H},deg={deg},seed={seed}): {N} Gaussians, {W}x{H}, SH deg {deg}, "
                               f"{'forward only' if a.forward_only else 'fwd+bwd'}, "
                               f"R={R} (tile,Gaussian) pairs, every view of a step within 0.1 % of that (cameras a few cm apart), "
                               f"frame-parallel dp{world}",
                   "gaussians": N, "width": W, "height": H, "sh_degree": deg, "num_rendered": R, "tile_list_mean": tile_mean,
                   "tile_list_max": tile_max, "views_per_step": k_views, "frames_per_launch": Kf, "launch_batches_per_step": n_batches,
                   "streams": n_streams, "regularisers": bool(a.regularisers),
                   "forward_only": bool(a.forward_only), "hip_graph": bool(a.graph),
                   "reduction": "one_shot" if a.one_shot_reduce or graph is not None else
                                f"fold of {rows} gradient row(s) for {k_views} views + collective in {len(batch.pipe.bounds)} chunk(s)",
                   "gradient_rows": rows,
                   "parallelism": f"dp{world}"},
        "roofline": roofline, "roofline_valu": roofline_valu,
        "roofline_one_view_per_step": {"achieved": total_bytes / (el_one / n_one) / 1e9, "unit": "GB/s",
                                       "frac": total_bytes / (el_one / n_one) / 1e9 / copy_gbs,
                                       "frac_of_spec": total_bytes / (el_one / n_one) / 1e9 / HBM_PEAK_GBS,
                                       "note": "the same whole-pass figure at the reference's one frame per step"},
        "hbm_copy_GBs_measured": copy_gbs,
        "kernel_ms": kern,
    }
    out.update(dinfo)
    out.update({k: None for k in COMM_KEYS})
    if comm is not None:
        out.update(comm)
    out["scaling_model"] = scaling_model(batch.acc.numel() * 4, ms_per_step, el_one / n_one * 1e3,
                                         comm.get("allreduce_exposed_ms") if comm else None)
    if grad_hash is not None:
        out["grad_sha256"] = grad_hash
    if dp_check is not None:
        out.update(dp_check)
    if one_by_algo is not None:
        out["one_view_per_step_by_algorithm"] = one_by_algo
    if world == 1 and not a.no_cpu_baseline:
        _log("CPU baseline (child process, bounded) + parity of the full-size views against the oracle")
        ins = (means3D, shs, opac, scales, rots)
        out["cpu_baseline"], out["parity"] = cpu_baseline(
            s, camera, deg, W, H, lambda v, dLn: _raster_view(eng, camera, v, ins, dLn, t, backward=not a.forward_only),
            n_views=1 if a.light else 3, lbs_project=not a.light, backward=not a.forward_only)
    return out


def _raster_view(eng, camera, v, ins, dLn, t, backward=True, host=True):
    """View v of the run's camera set through the one-view engine (forward + backward into its own gradient buffer), on the host:
    what the parity block and the frame-parallel self-check compare with the oracle / with the reduced sum."""
    import torch
    W, H, L = eng.W, eng.H, eng.L
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    eng.set_camera(camera(v)[3])
    eng._chain = None
    Rv = eng.forward(*ins, sync_num_rendered=True)
    if not 0 <= Rv <= eng.cap:
        if not host:
            raise SystemExit(f"bench.py: view {v}: R = {Rv} exceeds the engine's pair capacity {eng.cap}")
        return {"error": f"view {v}: R = {Rv} exceeds the engine's pair capacity {eng.cap}"}
    if backward:
        eng.backward(*ins, t(dLn))
    if not host:
        return {"R": Rv, "flat": eng.grad_flat}
    torch.cuda.synchronize()
    c = lambda x: x.detach().cpu().numpy()
    d = {"R": Rv, "radii": c(eng.radii), "color": c(eng.color),
         "ranges": c(eng.binning[L.bin_ranges:L.bin_ranges + 8 * Tn].view(torch.int32).view(Tn, 2)),
         "point_list": c(eng.binning[L.bin_point_list:L.bin_point_list + 4 * Rv].view(torch.int32))}
    if backward:
        d["grads"] = {"means3D": c(eng.d_means3D), "means2D": c(eng.d_means2D), "opacity": c(eng.d_opacity),
                      "scales": c(eng.d_scales), "rotations": c(eng.d_rots), "sh": c(eng.d_sh)}
    return d


def cpu_lbs_project_worker(argv):
    """Child process of the cpu_baseline leg (never touches the GPU): PyTorch-CPU "LBS + project" with `threads` threads,
    median of 10 runs at N = 6 890 / 50 k / 200 k (+ the workload's own N); prints one JSON line.  Runs in a child so that the
    parent can bound it with a timeout (an over-subscribed OpenMP team can take minutes per call).  `kind`: "raster" -- the first
    N Gaussians of the benchmark scene with seeded sparse J = 52 skinning weights and near-identity joint transforms (the
    arithmetic does not depend on their values); "avatar" -- the avatar scene's own canonical points, weights and an AMASS pose."""
    import math
    import numpy as np
    import torch
    from oracle import lbs_project_torch as lp
    threads, kind, Ntot, W, H, deg = int(argv[0]), argv[1], *(int(v) for v in argv[2:6])
    torch.set_num_threads(threads)
    T = torch.from_numpy
    J = 52
    if kind == "avatar":
        from oracle import lbs_oracle as lo
        from sings_amd.scene import avatar_scene
        s = avatar_scene(N=Ntot, J=J)
        cam = s["cam"]
        poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]
        pose = np.zeros(J * 3, np.float32); pose[:72] = poses72[0]; pose[:3] = 0
        R = lo.batch_rodrigues(T(pose).view(-1, 3)).view(1, J, 3, 3)
        A = lo.batch_rigid_transform(R, T(s["joints_rest"])[None], list(s["parents"]))[1][0]
        base = (s["xyz_canon"], s["scales"], s["opacities"], s["shs"], s["lbs_weights"])
        tail = (A, T(s["smpl_scale"]), T(s["transl"]), T(cam["world_view_transform"]), T(cam["full_proj_transform"]),
                T(cam["camera_center"]), s["W"], s["H"], math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5))
    else:
        from sings_amd.scene import synthetic_scene
        s = synthetic_scene(Ntot, W, H, deg, 3)
        rsd = np.random.RandomState(11)
        w = np.zeros((Ntot, J), np.float32)
        ja, jb = rsd.randint(0, J, Ntot), rsd.randint(0, J, Ntot)
        u = rsd.rand(Ntot).astype(np.float32)
        w[np.arange(Ntot), ja] = u; w[np.arange(Ntot), jb] += 1 - u
        A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1)); A[:, :3, 3] = rsd.normal(0, 1e-3, (J, 3))
        base = (s["means3D"], s["scales"], s["opacities"], s["shs"], w)
        tail = (T(A), torch.ones(1), torch.zeros(3), T(s["viewmatrix"]), T(s["projmatrix"]), T(s["campos"]), W, H, s["tanfovx"],
                s["tanfovy"])
    sweep = {}
    for n in dict.fromkeys((6890, 50000, 200000, Ntot)):
        idx = np.arange(n) % Ntot
        xyz, sc, op, sh, w_ = (T(np.ascontiguousarray(x[idx])) for x in base)
        args = (xyz, torch.eye(3)[None].repeat(n, 1, 1), sc, op, sh, deg, w_) + tail
        lp.lbs_project(*args)                                   # (first call: thread pool start-up)
        ts = []
        for _ in range(10):
            t1 = time.perf_counter(); lp.lbs_project(*args); ts.append(time.perf_counter() - t1)
        sweep[str(n)] = round(sorted(ts)[len(ts) // 2] * 1e3, 3)
    print(json.dumps({"threads": threads, "median_ms_by_points": sweep, "torch": torch.__version__}), flush=True)


def cpu_lbs_project(kind, Ntot, W, H, deg):
    """SURVEY.md 8(d) / BASELINE.md section 3: PyTorch-CPU "LBS + project" -- skinning (W.A, T[v;1]), rotation compose,
    matrix_to_quaternion, then cull / project / cov3D / cov2D / radius / SH -- on the host cores this process may use
    (`usable_cores`, stated), in a child process under a timeout; if the full team does not finish (over-subscription) the
    16-thread figure is reported and the line says so.  ONE implementation for every workload.  -> the cpu_baseline dict."""
    cores = usable_cores()
    tried, res = [], None
    for threads in dict.fromkeys((cores, min(cores, 16))):
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--_cpu-worker", str(threads), kind, str(Ntot), str(W),
                                str(H), str(deg)], capture_output=True, text=True, timeout=150)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and line:
                res = json.loads(line[-1])
                tried.append({"threads": threads, "ok": True})
                break
            tried.append({"threads": threads, "ok": False, "rc": p.returncode, "stderr": p.stderr[-300:]})
        except subprocess.TimeoutExpired:
            tried.append({"threads": threads, "ok": False, "timeout_s": 150})
    cpu_model = ""
    try:
        cpu_model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        pass
    if res is None:
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "port", "sample": "PyTorch-CPU LBS + project did not finish",
                "attempts": tried, "host_cpus": os.cpu_count(), "usable_cores": cores}
    ms = res["median_ms_by_points"][str(Ntot)] if str(Ntot) in res["median_ms_by_points"] else res["median_ms_by_points"]["200000"]
    n_at = Ntot if str(Ntot) in res["median_ms_by_points"] else 200000
    return {"value": 1e3 / ms, "unit": "frames/s (PyTorch-CPU LBS + project only: no binning, no composite, no backward)",
            "cores": res["threads"], "kind": "port",
            "sample": f"oracle/lbs_project_torch.py on the {kind} scene, J=52, SH deg {deg}, median of 10 runs per size, N={n_at}: {ms} ms",
            "median_ms_by_points": res["median_ms_by_points"], "cpu_model": cpu_model, "torch": res["torch"],
            "host_cpus": os.cpu_count(), "usable_cores": cores, "attempts": tried}


def cpu_baseline(s, camera, deg, W, H, gpu_view=None, n_views=3, lbs_project=True, backward=True):
    """The raster workloads' cpu_baseline + parity: `cpu_lbs_project` (the reported baseline), and the scalar C restatement of the
    whole rasterizer (1 core, `n_views` full views fwd+bwd of this run's cameras: a bounded sample) riding along as an extra key --
    its images and gradients are COMPARED with the engine's for the same cameras (`gpu_view(v, dL) -> dict`).
    -> (cpu_baseline, parity)."""
    from oracle import raster_oracle as ro
    Ntot = s["means3D"].shape[0]
    tc = 0.0
    par = _ParityLog()
    for v in range(n_views):
        v_, p_, c_, _ = camera(v)
        t0 = time.perf_counter()
        o = ro.forward(s["means3D"], s["opacities"], v_, p_, c_, W, H, s["tanfovx"], s["tanfovy"], s["bg"],
                       scales=s["scales"], rotations=s["rotations"], shs=s["shs"], sh_degree=deg, want_margin=True)
        # pixels whose hard-threshold decisions are borderline in the oracle carry no loss, on both sides (tests/test_gpu_raster.py)
        dLn = s["dL_dimage"].copy(); dLn[:, o["margin"] < PARITY_BORDER] = 0
        g = ro.backward(o, dLn) if backward else None
        tc += time.perf_counter() - t0
        if gpu_view is not None:                                 # the checker's result is USED: the engine's view v against it
            par.add(o, g, gpu_view(v, dLn))
    raster = {"value": n_views / tc, "unit": "views/s", "cores": 1, "kind": "port",
              "sample": f"{n_views} full view(s) {'fwd+bwd' if backward else 'forward'} of the same scene with the scalar C oracle "
                        f"({tc:.1f} s; the forward also computes the per-pixel threshold margins the parity block needs)"}
    parity = par.result() if gpu_view is not None else None
    if not lbs_project:
        return raster, parity
    cb = cpu_lbs_project("raster", Ntot, W, H, deg)
    cb["raster_oracle_1core"] = raster
    return cb, parity


PARITY_BORDER = 2e-5          # tests/test_gpu_raster.py::BORDER
PARITY_RGB_TOL = 1e-5         # BASELINE.json north_star: per-pixel RGB within 1e-5 of the reference


class _ParityLog:
    """HIP engine vs CPU oracle over the full-size views of the cpu_baseline leg -- the bars of tests/test_gpu_raster.py
    (bit-exact binning; RGB <= 1e-5 on pixels whose threshold decisions have a margin, borderline ones within the oracle's own
    flip bound; every gradient within rtol 2e-4 + 2e-6 of the array's scale), applied to the benchmark's own configuration on
    every run.  Outside the timed region; the oracle is the checker, never the thing measured as `value`."""

    def __init__(self):
        self.views = 0
        self.binning_exact = True
        self.rgb_linf = 0.0
        self.border_px = 0
        self.border_beyond_tol = 0
        self.border_beyond_flip = 0
        self.grad_max_rel = 0.0
        self.grad_violations = 0
        self.grads_compared = 0
        self.failed = []

    def add(self, o, g, d):
        import numpy as np
        self.views += 1
        if d.get("error"):
            self.failed.append(d["error"]); self.binning_exact = False
            return
        exact = (d["R"] == o["R"] and np.array_equal(d["radii"], o["radii"]) and
                 np.array_equal(d["ranges"].astype(np.uint32), o["ranges"]) and
                 np.array_equal(d["point_list"].astype(np.uint32), o["point_list"]))
        self.binning_exact = self.binning_exact and bool(exact)
        diff = np.abs(d["color"] - o["color"]).max(0)
        border = o["margin"] < PARITY_BORDER
        self.rgb_linf = max(self.rgb_linf, float(diff[~border].max()))
        self.border_px += int(border.sum())
        if border.any():
            self.border_beyond_tol += int((diff[border] > PARITY_RGB_TOL).sum())
            self.border_beyond_flip += int((diff[border] > PARITY_RGB_TOL + 1.001 * o["flip"][border]).sum())
        if g is None:
            return
        for name, key in (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmean2D"), ("opacity", "dL_dopacity"), ("scales", "dL_dscales"),
                          ("rotations", "dL_drots"), ("sh", "dL_dsh")):
            self.add_grad(d["grads"][name], g[key])

    def add_grad(self, a, b, rtol=2e-4, atol=2e-6):
        import numpy as np
        b = np.asarray(b, np.float64); a = np.asarray(a, np.float64).reshape(b.shape)
        scale = np.abs(b).max() + 1e-30
        err = np.abs(a - b)
        self.grads_compared += 1
        self.grad_max_rel = max(self.grad_max_rel, float(err.max() / scale))
        self.grad_violations += int((err > rtol * np.abs(b) + atol * scale).sum())

    def result(self):
        ok = (self.binning_exact and self.rgb_linf <= PARITY_RGB_TOL and self.border_beyond_flip == 0 and self.grad_violations == 0
              and not self.failed)
        return {"views": self.views, "ok": bool(ok), "binning_exact": bool(self.binning_exact), "rgb_linf": self.rgb_linf,
                "rgb_tol": PARITY_RGB_TOL, "borderline_px": self.border_px, "borderline_px_beyond_1e-5": self.border_beyond_tol,
                "borderline_px_beyond_flip_bound": self.border_beyond_flip, "grad_max_rel": self.grad_max_rel,
                "grad_violations": self.grad_violations, "gradient_arrays_compared": self.grads_compared, "grad_tol": "rtol 2e-4 + 2e-6 x max|g| per array (tests/test_gpu_raster.py)",
                "errors": self.failed,
                "against": "oracle/raster_oracle (scalar C restatement, fp32; PARITY UNPINNED: DESIGN.md section 2), full-size views "
                           "of this run's cameras 0..views-1, R / radii / ranges / point_list compared bit for bit"}


def _tile_list_stats(eng, W, H):
    """mean / max length of the per-tile depth-sorted lists of the engine's last forward (reported with
    every number); read from the tile ranges in the binning workspace."""
    import torch
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rg = eng.binning[eng.L.bin_ranges:eng.L.bin_ranges + 8 * T].view(torch.int32).view(T, 2)
    n = (rg[:, 1] - rg[:, 0]).clamp_(min=0)
    return float(n.float().mean().item()), int(n.max().item())


def _emit(obj):
    """Print the result as the LAST line of stdout: text that native libraries (RCCL) left in the C stdio buffer is
    flushed first, otherwise it would come out at process exit, after the JSON line."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)


def leg_train(a, ctx):
    """Extra workload: ONE COMPLETE training step of an avatar through autograd -- tri-plane + decoder decode of all
    Gaussians, fused LBS + raster forward, clamp + L1 + SSIM, L2Norm + Gaussian edge regularisers, backward through all
    of it to the planes / decoder weights / anchors (SURVEY.md 3.1 without optimiser and densification)."""
    import math
    import numpy as np
    import torch
    rank, world, dev, dist, dinfo = ctx
    from sings_amd.body import joint_transforms
    from sings_amd.decode import (AppearanceDecoder, GeometryDecoder, HexPlaneField, arena_sync, overlap_weight_grads,
                                  prepare_triplane_backward_early, set_gradient_arena)
    overlap_weight_grads(not a.no_wgrad_overlap)     # (the step sets every .grad to None first: the mode's precondition)
    prepare_triplane_backward_early(not a.no_wgrad_overlap)   # (every captured forward is followed by its backward)
    from sings_amd.dp import FrameSharder
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.regularizers import GaussiansEdgeLoss, L2Norm
    from sings_amd.scene import avatar_scene
    from sings_amd.train_step import AvatarStep
    N = a.gaussians if a.gaussians != 200000 else 150000
    s = avatar_scene(N=N, J=52)
    W, H, J = s["W"], s["H"], s["J"]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    cam = s["cam"]
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5), tanfovy=math.tan(cam["fovy"] * 0.5), bg=t(s["bg"]),
        scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]), projmatrix=t(cam["full_proj_transform"]), sh_degree=0,
        campos=t(cam["camera_center"]), prefiltered=False, debug=False)
    poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]
    F = poses72.shape[0]
    poses = np.zeros((F, J * 3), np.float32); poses[:, :72] = poses72; poses[:, :3] = 0
    jr = t(s["joints_rest"])
    A_all = torch.stack([joint_transforms(t(poses[f]), jr, tuple(s["parents"])) for f in range(F)]).reshape(F, J, 4, 4).contiguous()
    torch.manual_seed(0)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [64, 64, 64],
           'multires': [1, 2, 4]}                                              # human_complex.yaml:38-43
    # (feature-minor planes: the reference's shapes and state_dict, channels_last in memory -- used in place by the sampling kernels)
    tri = HexPlaneField(cfg, bounds=1.2, device=dev, feature_minor=not os.environ.get("SINGS_PLANES_NCHW"))
    geo = GeometryDecoder(96).to(dev); app = AppearanceDecoder(96).to(dev)
    with torch.no_grad():                                                      # millimetre-sized splats, tiny offsets
        geo.scales[2].bias.fill_(-5.3); geo.scales[2].weight.mul_(0.1)
        geo.xyz_offsets.weight.mul_(0.01); geo.xyz_offsets.bias.zero_()
    step_mod = AvatarStep(t(s["xyz_canon"]), t(s["lbs_weights"]), tri, geo, app, l2_norm=L2Norm(),
                          gaussian_connect=GaussiansEdgeLoss(), gaussian_connect_w=1.0,
                          defer_regulariser_join=not a.join_regularisers_early).to(dev)
    params = [p for p in step_mod.parameters() if p.requires_grad]
    gt_rgb = torch.rand((3, H, W), device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    mask = ((((xx - W / 2) / (W / 4)) ** 2 + ((yy - H / 2) / (H / 2.2)) ** 2) < 1).float().contiguous()
    bg_t, smpl_scale, transl = t(s["bg"]), t(s["smpl_scale"]), t(s["transl"])
    shard = FrameSharder(F, world, rank, seed=0)

    # (before the first backward and before the capture: the gradient arena decides where the large gradients are WRITTEN)
    fp = None
    if dist is not None:
        from sings_amd.dp import FrameParallel
        fp, algo_info = make_frame_parallel(ctx, sum(p.numel() for p in params))
        # parameter-level gradients live in ONE flat buffer: the kernels that produce the large ones (tri-plane scatter, weight
        # gradients) write straight into it (sings_amd.decode.set_gradient_arena), so p.grad is a view of `flat` and the
        # collective needs no gather / scatter passes; what torch's own backward produced (biases, anchors: < 1 %) is copied
        flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        grad_views = set_gradient_arena(params, flat)

    # No host synchronisation inside a step: synchronous steps size the pair capacity, then the rasterizer's pair-count
    # check is deferred (sings_amd.rasterizer.set_deferred_overflow_check) and polled once after the timed region.
    from sings_amd import rasterizer as _rz
    # --views-per-step K > 1: a CHUNK of K frames per optimisation step (AvatarStep with A [K,J,4,4]): one decode, K frames rendered,
    # compared and differentiated in one call per direction; the default 1 is the reference's step
    Kt = max(1, min(int(a.views_per_step), 16))
    A_static = A_all[0].clone() if Kt == 1 else A_all[:Kt].clone()

    def step_body():
        for p in params:
            p.grad = None
        loss, ld, ex = step_mod(A_static, rs, gt_rgb, mask, bg_t, smpl_scale=smpl_scale, transl=transl)
        if loss is None:                                            # two roots: the regularisers join where their gradient is consumed
            step_mod.backward(ld, ex)
        else:
            loss.backward()
        keys = list(ld.keys())                                      # (no autograd graph kept alive across the end of a capture;
        vals = torch.stack([ld[k].detach().reshape(()) for k in keys])   # one launch for all the scalars)
        return {k: vals[i] for i, k in enumerate(keys)}

    cap_pairs = 0
    _rz.set_overflow_check("sync")                               # every sizing step reads its pair count before it returns, so
    for f in range(0, F, 8):                                     # _capacity_hint (2 x the largest count) has seen them all
        A_static.copy_(A_all[f] if Kt == 1 else A_all[torch.arange(f, f + Kt, device=dev) % F])
        step_body()
        cap_pairs = max(cap_pairs, _rz._capacity_hint[dev.index])
    torch.cuda.synchronize()
    _rz.set_deferred_overflow_check(True, capacity_pairs=cap_pairs)
    if os.environ.get("SINGS_TORCH_PROFILE"):                    # which torch ops (copies, additions) sit between the library's kernels
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            for _ in range(2):
                step_body()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=120, max_name_column_width=50,
                                                                 max_shapes_column_width=60), file=sys.stderr)
        # every aten op that launches something, with the first frame of this repository on its stack
        seen = {}
        for ev in prof.events():
            dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0) or 0
            if ev.name.startswith("aten::") and dt > 0 and not any(c.name.startswith("aten::") for c in ev.cpu_children):
                fr = next((f for f in (ev.stack or []) if "/sings_amd/" in f or "bench.py" in f), "?")
                k = (ev.name, str(ev.input_shapes)[:60], fr.strip()[-90:])
                c = seen.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += dt
        for (n, sh, fr), (c, t) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
            print(f"ATEN {n:22s} x{c:3d} {t:8.1f} us  {sh:60s} {fr}", file=sys.stderr)
    graph, ld_static = None, None
    if not a.eager:
        # the whole step (decode -> raster -> losses -> backward, ~600 launches) replayed from ONE HIP graph
        # (capture_step: warm-up on a side stream, and an error instead of a dead process if the step handed back tensors that
        #  still carry their autograd graph -- hipStreamEndCapture segfaults on those, round 3)
        from sings_amd.train_step import capture_step
        graph, ld_static = capture_step(step_body, warmup=3, device=dev)

    frame_ix = torch.empty(Kt, dtype=torch.long, device=dev)
    pins = [torch.empty(Kt, dtype=torch.long).pin_memory() for _ in range(64)]

    def step(i):
        if Kt == 1:
            A_static.copy_(A_all[shard.frame(i)])
        else:                                                    # the step's K frames: gathered on the device (no host wait)
            pin = pins[i % 64]
            pin.copy_(torch.tensor([shard.frame(i * Kt + k) for k in range(Kt)], dtype=torch.long))
            frame_ix.copy_(pin, non_blocking=True)
            torch.index_select(A_all, 0, frame_ix, out=A_static)
        if graph is not None:
            graph.replay()
            ld = ld_static
        else:
            ld = step_body()
        if fp is not None:
            arena_sync(params, grad_views, True)
            fp.all_reduce_grads(flat)
            arena_sync(params, grad_views, False)
        return ld

    for i in range(a.warmup):
        step(i)
    last = {}

    def timed_step(i):
        last["ld"] = step(a.warmup + i)
    els = timed_repeats(dist, dev, a.steps, timed_step, min_s=LIGHT_TIMED_S if a.light else None)
    el = _median(els)
    ld = last["ld"]
    comm = allreduce_probe(fp, flat) if fp is not None else None
    inplace = (sum(p.grad.numel() * 4 for p, v in zip(params, grad_views) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
               if fp is not None else None)
    R_last = _rz.check_deferred_overflow(dev)                    # raises if a timed step overflowed the pair capacity
    if rank == 0:
        nparam = sum(p.numel() for p in params)
        out = ({
            "metric": "full train-step views/sec (decode + LBS-fused raster + L1/SSIM + regularisers, fwd+bwd), avatar ~150k Gaussians",
            "value": world * Kt * a.steps / el, "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "ms_per_view": el / a.steps / Kt * 1e3, "timed_region_s": sum(els), "repeats": len(els),
            "ms_per_step_min": min(els) / a.steps * 1e3, "ms_per_step_max": max(els) / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"avatar_scene(N={N}, J={J}) {W}x{H}, tri-plane 32 x (64,128,256)^2 x 3, decoders 96-128-128 / "
                                   f"96-64-64, SH deg 0, {F} AMASS frames, no optimiser step, frame-parallel dp{world}",
                       "gaussians": N, "frames_per_step": Kt, "trainable_parameters": nparam, "num_rendered_last": R_last, "hip_graph": not a.eager,
                       "gradient_bytes_written_in_place": inplace,
                       "parallelism": f"dp{world}"},
            "losses": {k: float(v.detach()) for k, v in ld.items()}})
        out.update(dinfo)
        out.update({k: None for k in COMM_KEYS})
        if comm is not None:
            out.update(comm)
        if fp is not None:
            out.update(algo_info)
        out["schedule_note"] = ("one optimisation step per timed step: at frames_per_step = 1 this is the reference's batch-1 schedule "
                                "(config.py:27, gs_trainer.py:209-254); at K > 1 ONE optimiser step consumes a chunk of K frames -- "
                                "K times fewer parameter updates per frame, a different optimisation trajectory, quoted per frame "
                                "for throughput only")
        copy_gbs = a.copy_gbs if getattr(a, "copy_gbs", None) else measure_copy_peak(dev)
        out["hbm_copy_GBs_measured"] = copy_gbs
        out["roofline"] = train_step_roofline(N, Kt, (tri, geo, app), W, H, int(R_last), J, el / a.steps * 1e3, copy_gbs)
        if world == 1 and not a.no_cpu_baseline:
            _log("train: parity of one step's decode / image / loss against the oracle chain")
            out["parity"] = train_parity(s, step_mod, (tri, geo, app), rs, A_all[0], gt_rgb, mask, bg_t, smpl_scale, transl)
    _rz.set_overflow_check("sync")
    return out if rank == 0 else None


def train_parity(s, step_mod, mods, rs, A, gt_rgb, mask, bg_t, smpl_scale, transl):
    """ONE forward of the complete training step against the chain of CPU oracles (the forward half of
    tests/test_gpu_train_step.py::test_full_step_matches_oracle_chain, at the benchmark's full size): decoded attributes vs
    oracle/decode_oracle.py (pinned by the reference-generated decode_golden.npz), the image vs lbs_oracle -> raster_oracle on
    those attributes, the L1 / SSIM loss values vs oracle/photo_loss_oracle.py (pinned by photo_loss_golden.npz)."""
    import math
    import numpy as np
    import torch
    from oracle import decode_oracle as do
    from oracle import lbs_oracle as lo
    from oracle import photo_loss_oracle as plo
    from oracle import raster_oracle as ro
    tri, geo, app = mods
    with torch.no_grad():
        loss, ld, ex = step_mod(A, rs, gt_rgb, mask, bg_t, smpl_scale=smpl_scale, transl=transl)
    torch.cuda.synchronize()
    c = lambda x: x.detach().float().cpu()
    N = int(step_mod.xyz.shape[0])
    with torch.no_grad():
        grids = [[c(p_) for p_ in gp] for gp in tri.grids]
        feats = do.triplane_features(c(step_mod.xyz), grids, c(tri.aabb))
        og = do.geometry_decoder(feats, {k: c(v) for k, v in geo.named_parameters()})
        oa = do.appearance_decoder(feats, {k: c(v) for k, v in app.named_parameters()})
        want = {"xyz_canon": c(step_mod.xyz) + og["xyz_offsets"], "scales": og["scales"], "opacity": oa["opacity"], "shs": oa["shs"]}
    at = ex["attrs"]
    dec_rel, dec_bad = 0.0, 0
    for k, b in want.items():
        a_ = c(at[k]).numpy().astype(np.float64).reshape(b.shape); b = b.numpy().astype(np.float64)
        scale = np.abs(b).max() + 1e-30
        err = np.abs(a_ - b)
        dec_rel = max(dec_rel, float(err.max() / scale))
        dec_bad += int((err > 1e-4 * np.abs(b) + 1e-5 * scale).sum())
    # raster + loss on the GPU's OWN decoded attributes (the seam to the decode oracle is the comparison above)
    cam = s["cam"]
    A_n = c(A).reshape(-1, 4, 4)
    pxyz, pq, psc, _ = lo.deform_gaussians(c(at["xyz_canon"]), torch.eye(3)[None].repeat(N, 1, 1), c(at["scales"]),
                                           torch.from_numpy(s["lbs_weights"]), A_n, smpl_scale=c(smpl_scale), transl=c(transl))
    o = ro.forward(pxyz.numpy(), c(at["opacity"]).numpy(), cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                   s["W"], s["H"], math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), s["bg"], scales=psc.numpy(),
                   rotations=pq.numpy(), shs=c(at["shs"]).numpy(), sh_degree=0)
    img = c(ex["render_raw"]).numpy()
    diff = np.abs(img - o["color"]).max(0)
    border = o["margin"] < PARITY_BORDER
    # (posed by the LBS ORACLE here, not by the kernel: a last-ulp difference of a posed quaternion may move a splat's rectangle --
    #  such pixels are counted, not excused)
    pl = plo.photometric_loss(torch.from_numpy(o["color"]), c(gt_rgb), c(mask), c(bg_t), 0.8, 0.2)
    l1_rel = abs(float(ld["l1"]) - float(pl["l1"])) / max(abs(float(pl["l1"])), 1e-30)
    ssim_rel = abs(float(ld["ssim"]) - float(pl["ssim"])) / max(abs(float(pl["ssim"])), 1e-30)
    over = int((diff[~border] > PARITY_RGB_TOL).sum())
    ok = dec_bad == 0 and over <= 1e-5 * diff.size and l1_rel <= 2e-5 and ssim_rel <= 2e-5
    return {"ok": bool(ok), "decode_max_rel": dec_rel, "decode_violations": dec_bad,
            "decode_tol": "rtol 1e-4 + 1e-5 x max|x| per attribute (tests/test_gpu_decode.py)", "num_rendered_oracle": int(o["R"]),
            "rgb_linf_median_px": float(np.median(diff)), "rgb_px_beyond_1e-5": over, "borderline_px": int(border.sum()),
            "l1_rel_err": l1_rel, "ssim_rel_err": ssim_rel, "loss_tol": 2e-5,
            "against": "oracle/decode_oracle -> lbs_oracle -> raster_oracle (PARITY UNPINNED) -> photo_loss_oracle, one full-size "
                       "forward of this run's step (frame 0); gradients of the composed step: tests/test_gpu_train_step.py"}


def leg_avatar(a, ctx):
    """BASELINE configs[3]: frame-parallel training step of an avatar through the LBS-fused kernels."""
    import math
    import numpy as np
    import torch
    rank, world, dev, dist, dinfo = ctx
    from sings_amd import _lib
    from sings_amd.body import joint_transforms
    from sings_amd.dp import FrameParallel, FrameSharder
    from sings_amd.engine import SkinnedEngine
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import avatar_scene
    N = a.gaussians if a.gaussians != 200000 else 150000
    s = avatar_scene(N=N, J=52)
    if a.morton:
        from sings_amd.scene import morton_order
        perm = morton_order(s["xyz_canon"])
        for key in ("xyz_canon", "lbs_weights", "scales", "opacities", "shs"):
            s[key] = np.ascontiguousarray(s[key][perm])
    W, H, J = s["W"], s["H"], s["J"]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    cam = s["cam"]
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5), tanfovy=math.tan(cam["fovy"] * 0.5), bg=t(s["bg"]),
        scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]), projmatrix=t(cam["full_proj_transform"]), sh_degree=0,
        campos=t(cam["camera_center"]), prefiltered=False, debug=False)          # human.sh_degree: 0 (human_complex.yaml:34)
    poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]      # [120,72] AMASS frames
    F = poses72.shape[0]
    poses = np.zeros((F, J * 3), np.float32); poses[:, :72] = poses72
    poses[:, :3] = 0                                                             # global orient: face the camera
    jr = t(s["joints_rest"])
    A_all = torch.stack([joint_transforms(t(poses[f]), jr, tuple(s["parents"])) for f in range(F)]).reshape(F, J, 16).contiguous()
    xyz, w, sc, op, sh = t(s["xyz_canon"]), t(s["lbs_weights"]), t(s["scales"]), t(s["opacities"]), t(s["shs"])
    smpl_scale, transl, dL = t(s["smpl_scale"]), t(s["transl"]), t(s["dL_dimage"])
    eng = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=16 * N + 65536)
    eng.set_camera(rs)
    Rmax, tile_mean, tile_max = 0, 0.0, 0
    for f in range(0, F, 8):
        eng.set_frame(xyz, None, w, A_all[f], smpl_scale, transl)
        Rf = eng.forward(sh, op, sc, sync_num_rendered=True)
        if Rf > Rmax:
            Rmax = Rf
            tile_mean, tile_max = _tile_list_stats(eng, W, H)
    del eng
    torch.cuda.empty_cache()
    # one engine (workspaces) + loss engine per view of the batch, each writing its own row of `grads`; the views are dealt
    # round-robin to the streams; one pass sums the rows, one all-reduce per step (same scheme as the raster workload)
    k_views = max(1, a.views_per_step)
    # K frames per launch (round 4): the step's k_views frames go out as k_views / K batches, ONE dispatch per kernel and batch
    Kf = max(1, min(a.frames_per_launch if a.frames_per_launch is not None else 8, k_views, _lib.MAX_FRAMES))
    while k_views % Kf:
        Kf -= 1
    n_batches = k_views // Kf
    n_streams = max(1, min(a.streams, n_batches))
    per_view = N * (3 + 3 + 1 + 3 * sh.shape[1])
    from sings_amd.engine import SkinnedFramesEngine, ViewBatch
    from sings_amd.photo_loss import PhotoLossEngine
    pipelined = Kf > 1 and n_batches > 1 and a.pipeline
    rows = 1 if pipelined else {"streams": n_streams, "views": n_batches, "one": 1}[a.gradient_rows]
    # SH gradients coefficient-major (SG_FLAG_SH_PLANAR): the reference allocates 16 SH rows and trains at degree 0 -- 45 of the 55
    # gradient floats per Gaussian are structural zeros.  Only the prefix that carries gradient (10 floats per Gaussian) is
    # written, folded and all-reduced; the rest of the buffer is zeroed once, here
    grads = torch.zeros((rows, per_view), dtype=torch.float32, device=dev)
    engs, losses = [], []
    for v in range(n_batches):
        if Kf == 1:
            e = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=int(Rmax * 1.3) + 4096, grad_flat=grads[v % rows], sh_planar=True)
        else:
            e = SkinnedFramesEngine(N, J, W, H, sh.shape[1], Kf, dev, capacity_pairs=int(Rmax * 1.3) + 4096, grad_flat=grads[v % rows],
                                    sh_planar=True)
        e.set_camera(rs)
        engs.append(e)
        losses.append(PhotoLossEngine(W, H, dev, l1_w=0.8, ssim_w=0.2, K=Kf))    # human.loss.l1_w / ssim_w
    eng = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=int(Rmax * 1.3) + 4096, sh_planar=True) if Kf > 1 else engs[0]
    active = eng.active_floats(0)                                    # N * 10 of the N * 55 floats
    if Kf > 1:
        eng.set_camera(rs)
    loss1 = PhotoLossEngine(W, H, dev, l1_w=0.8, ssim_w=0.2) if Kf > 1 else losses[0]
    transl_k = transl[None].repeat(Kf, 1).contiguous()                            # (per-frame translations: here all equal)
    shard = FrameSharder(F, world, rank, seed=0)
    fp, algo_info = make_frame_parallel(ctx, active)
    batch = ViewBatch(engs, grads, n_streams, frame_parallel=fp, chunks=a.reduce_chunks, active=active)
    # train step = fused LBS+raster forward -> clamp + L1 + SSIM loss against a (random) target with a body-shaped
    # mask, forward and gradient -> backward (SURVEY.md 8d "Timing")
    torch.manual_seed(0)                                             # (the target image: the same in every process)
    gt_rgb = torch.rand((3, H, W), device=dev)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    mask = ((((xx - W / 2) / (W / 4)) ** 2 + ((yy - H / 2) / (H / 2.2)) ** 2) < 1).float().contiguous()
    bg_t = t(s["bg"])

    def one_view(v, frame, e=None, le=None):
        e = engs[v] if e is None else e
        e.set_frame(xyz, None, w, A_all[frame], smpl_scale, transl)
        e.forward(sh, op, sc)
        dLi = (losses[v] if le is None else le)(e.color, gt_rgb, mask, bg_t)
        e.backward(sh, op, sc, dLi)

    frame_idx = [torch.empty(Kf, dtype=torch.long, device=dev) for _ in range(n_batches)]
    A_batch = [torch.empty((Kf, J, 16), dtype=torch.float32, device=dev) for _ in range(n_batches)]
    # the frame numbers of a batch reach the device through a ring of pinned host words (an asynchronous 64-byte copy: the host
    # never waits, and runs at most a few steps ahead of the device -- the ring is 256 steps deep)
    pins = [[torch.empty(Kf, dtype=torch.long).pin_memory() for _ in range(256)] for _ in range(n_batches)]
    pin_at = [0] * n_batches

    def one_batch(b, frames):
        """Kf frames of the step in ONE dispatch per kernel: their joint transforms gathered into [Kf,J,16] (one small launch)."""
        e = engs[b]
        pin = pins[b][pin_at[b] % 256]; pin_at[b] += 1
        pin.copy_(torch.tensor(frames, dtype=torch.long))
        frame_idx[b].copy_(pin, non_blocking=True)
        torch.index_select(A_all, 0, frame_idx[b], out=A_batch[b])
        e.set_frames(xyz, None, w, A_batch[b], smpl_scale, transl_k)
        e.forward(sh, op, sc)
        dLi = losses[b](e.color, gt_rgb, mask, bg_t)                            # (one target image for all frames: stride 0)
        e.backward(sh, op, sc, dLi)

    if pipelined:
        # ONE gradient buffer for the step (rows = 1): the batches' per-Gaussian halves run in batch order on one stream
        from sings_amd.engine import FramePipeline
        pipe2 = FramePipeline(engs, dev)

        def set_frames(b, frames):
            pin = pins[b][pin_at[b] % 256]; pin_at[b] += 1
            pin.copy_(torch.tensor(frames, dtype=torch.long))
            frame_idx[b].copy_(pin, non_blocking=True)
            torch.index_select(A_all, 0, frame_idx[b], out=A_batch[b])
            engs[b].set_frames(xyz, None, w, A_batch[b], smpl_scale, transl_k)

    def step(i):
        if Kf == 1:
            batch.run(lambda v, e: one_view(v, shard.frame(i * k_views + v)))
        elif pipelined:
            pipe2.run(prepare=lambda b, e: set_frames(b, [shard.frame(i * k_views + b * Kf + f) for f in range(Kf)]),
                      forward_args=(sh, op, sc),
                      loss=lambda b, e: losses[b](e.color, gt_rgb, mask, bg_t))
            batch.pipe.reduce()                                  # (one row: no fold; with several ranks the all-reduce)
        else:
            batch.run(lambda b, e: one_batch(b, [shard.frame(i * k_views + b * Kf + f) for f in range(Kf)]))

    for i in range(a.warmup):
        step(i)
    els = timed_repeats(dist, dev, a.steps, lambda i: step(a.warmup + i), min_s=LIGHT_TIMED_S if a.light else None)
    el = _median(els)
    comm = allreduce_probe(fp, batch.acc[:active])
    assert all(max(e.num_rendered()) <= e.cap if Kf > 1 else e.num_rendered() <= e.cap for e in engs)
    grad_hash = None
    if a.grad_hash:
        step(0)
        torch.cuda.synchronize()
        grad_hash = _grad_sha256(batch.acc)
    # the reference's unit of work: ONE frame per optimisation step on the current stream (gs_trainer.py:207-215)
    def step_one_frame(i):
        one_view(0, shard.frame(i), eng, loss1)
        if fp is not None:
            fp.all_reduce_grads(eng.grad_flat[:active])
    n_one = max(20, min(a.steps * k_views, 2000))
    eng.throughput = False                                       # one frame in flight from here on (SG_FLAG_THROUGHPUT off)
    for i in range(10):
        step_one_frame(i)
    el_one = _median(timed_repeats(dist, dev, n_one, step_one_frame, min_s=0.25))
    lib = _lib.load()
    lib.sg_profile_enable(1)
    for i in range(a.steps):
        one_view(0, shard.frame(i), eng, loss1)
    ms = (C.c_double * _lib.NUM_KERNELS)(); cnt = (C.c_int64 * _lib.NUM_KERNELS)()
    _lib.check(lib.sg_profile_collect(ms, cnt, _lib.NUM_KERNELS), "profile")
    lib.sg_profile_enable(0)
    kern = {lib.sg_kernel_name(k).decode(): (ms[k] / max(cnt[k], 1)) for k in range(_lib.NUM_KERNELS)}
    dp_check = None
    if dist is not None and not pipelined:
        def render_flat(v):
            r_, j_ = divmod(v, k_views)                              # view j of rank r at step 0
            one_view(0, FrameSharder(F, world, r_, seed=0).frame(j_), eng, loss1)
            return eng.grad_flat[:active]
        dp_check = dp_self_check(ctx, lambda: (step(0), batch.acc[:active])[1], render_flat, world * k_views)
        dp_check.update(algo_info)
        dp_check.update(densification_stats_check(ctx, fp, engs))
    if rank == 0:
        out = {"metric": "train-step views/sec (LBS-fused fwd + L1/SSIM loss + bwd), avatar ~150k Gaussians x 120 AMASS frames",
               "value": world * a.steps * k_views / el, "unit": "views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": el / a.steps * 1e3, "ms_per_view": el / a.steps * 1e3 / k_views,
               "train_step_ms_one_view": el_one / n_one * 1e3, "timed_region_s": sum(els), "repeats": len(els),
               "ms_per_step_min": min(els) / a.steps * 1e3, "ms_per_step_max": max(els) / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"avatar_scene(N={N}, J={J}) {W}x{H} fx=fy=5000, {F} AMASS frames, SH deg 0, fused LBS+raster "
                                      f"fwd + L1/SSIM loss + bwd, R<={Rmax}, frame-parallel dp{world}", "gaussians": N, "joints": J,
                          "width": W, "height": H, "max_num_rendered": Rmax, "tile_list_mean": tile_mean, "tile_list_max": tile_max,
                          "views_per_step": k_views, "frames_per_launch": Kf, "launch_batches_per_step": n_batches,
                          "streams": 2 if pipelined else n_streams,
                          "schedule": "pipeline: composite kernels on one stream, binning / loss / per-Gaussian backward of the "
                                      "other batches beside them on a high-priority stream" if pipelined else
                                      "every batch runs its whole chain on one of the streams",
                          "parallelism": f"dp{world}"},
               "kernel_ms": kern}
        per, total_bytes = algorithmic_bytes_skinned(N, H, W, Rmax, 0, J)
        fps = out["value"] / world
        copy_gbs = a.copy_gbs if getattr(a, "copy_gbs", None) else measure_copy_peak(dev)
        out["roofline"], out["roofline_valu"] = build_roofline(
            kern, per, {"workload": "avatar", "gaussians": N, "width": W, "height": H, "sh_degree": 0}, total_bytes, 1.0 / fps,
            copy_gbs, frames=Kf)
        out["roofline"]["note"] = ("raster + fused LBS bytes at the largest R of the sequence; the L1 + SSIM loss inside the timed "
                                   "step (HW 40 B algorithmic) is not counted")
        out["hbm_copy_GBs_measured"] = copy_gbs
        out.update(dinfo)
        out.update({k: None for k in COMM_KEYS})
        if comm is not None:
            out.update(comm)
        out["scaling_model"] = scaling_model(active * 4, el / a.steps * 1e3, el_one / n_one * 1e3,
                                             comm.get("allreduce_exposed_ms") if comm else None)
        out["gradient_floats_per_gaussian"] = {"buffer": per_view // N, "carrying_gradient": active // N,
                                               "note": "SH gradients coefficient-major (SG_FLAG_SH_PLANAR): only the (sh_degree+1)^2 "
                                                       "planes in use are written, folded and all-reduced"}
        if grad_hash is not None:
            out["grad_sha256"] = grad_hash
        if world == 1 and not a.no_cpu_baseline:
            _log("avatar: parity of one full-size frame against the composed oracle" + ("" if a.light else " + CPU baseline"))
            out["parity"] = avatar_parity(s, eng, rs, A_all[shard.frame(0)], (xyz, w, sc, op, sh, smpl_scale, transl), t)
            if not a.light:
                out["cpu_baseline"] = cpu_lbs_project("avatar", N, W, H, 0)
        if dp_check is not None:
            out.update(dp_check)
    return out if rank == 0 else None


def avatar_parity(s, eng, rs, A, ins, t):
    """ONE full-size frame of the avatar workload against the composed oracle (the bars of tests/test_gpu_skinned.py::
    test_cfg4_full_size_against_the_oracle): posed values vs oracle/lbs_oracle.py in ulps, the raster oracle on the kernel's own
    posed values (R, radii, ranges, sorted lists bit for bit; RGB <= 1e-5 off borderline pixels), and the gradients w.r.t. the
    canonical means / scales / opacity / SH, dL/dA, dL/dtransl and the screen-space statistic vs the raster oracle's explicit
    backward chained through the LBS oracle's autograd.  Outside every timed region; the oracle is the checker only."""
    import math
    import numpy as np
    import torch
    from oracle import lbs_oracle as lo
    from oracle import raster_oracle as ro
    xyz, w, sc, op, sh, smpl_scale, transl = ins
    N, J, W, H, dev = eng.P, eng.J, eng.W, eng.H, eng.dev
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    posed = (e(N, 3), e(N, 4), e(N, 3))
    eng.set_camera(rs)
    eng.set_frame(xyz, None, w, A, smpl_scale, transl)
    eng._chain = None
    Rv = eng.forward(sh, op, sc, sync_num_rendered=True, posed_out=posed)
    torch.cuda.synchronize()
    c = lambda x: x.detach().cpu().numpy()
    pxyz, pq, psc = (c(x) for x in posed)
    cam = s["cam"]
    o = ro.forward(pxyz, s["opacities"], cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"], W, H,
                   math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), s["bg"], scales=psc, rotations=pq, shs=s["shs"], sh_degree=0)
    dLn = s["dL_dimage"].copy(); dLn[:, o["margin"] < PARITY_BORDER] = 0
    g = ro.backward(o, dLn)
    eng.backward(sh, op, sc, t(dLn))
    torch.cuda.synchronize()
    L, Tn = eng.L, ((W + 15) // 16) * ((H + 15) // 16)
    par = _ParityLog()
    par.add(o, None, {"R": Rv, "radii": c(eng.radii), "color": c(eng.color),
                      "ranges": c(eng.binning[L.bin_ranges:L.bin_ranges + 8 * Tn].view(torch.int32).view(Tn, 2)),
                      "point_list": c(eng.binning[L.bin_point_list:L.bin_point_list + 4 * max(Rv, 0)].view(torch.int32))})
    # LBS^T by the oracle's autograd, seeded with the raster oracle's posed-space gradients
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True)
    xo, so, Ao, to = T(s["xyz_canon"]), T(s["scales"]), T(c(A).reshape(J, 4, 4)), T(s["transl"])
    px, pqo, pso, _ = lo.deform_gaussians(xo, torch.eye(3)[None].repeat(N, 1, 1), so, torch.from_numpy(s["lbs_weights"]), Ao,
                                          smpl_scale=torch.from_numpy(s["smpl_scale"]), transl=to)
    ((px * torch.from_numpy(g["dL_dmeans3D"])).sum() + (pqo * torch.from_numpy(g["dL_drots"])).sum()
     + (pso * torch.from_numpy(g["dL_dscales"])).sum()).backward()
    mag = np.abs(px.detach().numpy()).max(1, keepdims=True)
    ulps = {"means": float((np.abs(pxyz.astype(np.float64) - px.detach().numpy()) / np.spacing(mag.astype(np.float32))).max()),
            "quaternions_of_1": float((np.abs(pq.astype(np.float64) - pqo.detach().numpy()) / np.spacing(np.float32(1))).max()),
            "scales": float((np.abs(psc.astype(np.float64) - pso.detach().numpy()) / np.spacing(np.abs(pso.detach().numpy()))).max())}
    # (segmented backward of long lists: the tolerances of the full-size test)
    dsh = c(eng.d_sh)
    dsh = dsh.transpose(1, 0, 2) if eng.sh_planar else dsh                   # [M,P,3] -> [P,M,3]
    for a_, b_, rt, at in ((c(eng.d_xyz), xo.grad.numpy(), 1e-3, 1e-5), (c(eng.d_scales), so.grad.numpy(), 1e-3, 1e-5),
                           (c(eng.d_opacity), g["dL_dopacity"], 1e-3, 1e-5), (dsh[:, :1], g["dL_dsh"][:, :1], 1e-3, 1e-5),
                           (c(eng.d_means2D), g["dL_dmean2D"], 1e-3, 1e-5),
                           (c(eng.d_A).reshape(J, 4, 4)[:, :3], Ao.grad.numpy()[:, :3], 2e-3, 2e-4), (c(eng.d_transl), to.grad.numpy(), 2e-3, 2e-4)):
        par.add_grad(a_, b_, rtol=rt, atol=at)
    res = par.result()
    res["posed_ulps_vs_lbs_oracle"] = ulps
    res["ok"] = bool(res["ok"] and ulps["means"] <= 2 and ulps["quaternions_of_1"] <= 8 and ulps["scales"] <= 1)
    res["grad_tol"] = "rtol 1e-3 + 1e-5 x max|g| (dL/dA, dL/dtransl: 2e-3 + 2e-4): tests/test_gpu_skinned.py, full-size avatar"
    res["against"] = ("oracle/lbs_oracle (pinned by the reference-generated lbs_golden.npz) composed with oracle/raster_oracle (PARITY "
                      "UNPINNED) on ONE full-size frame of this run: posed values in ulps, R / radii / ranges / point_list bit for "
                      "bit, image, 7 gradient arrays incl. dL/dA and dL/dtransl")
    return res


def leg_dropin(a, ctx):
    """The drop-in autograd surface an UNMODIFIED gs_renderer_single.render() calls (GaussianRasterizer.forward / backward through
    torch autograd, default overflow mode: the pair count is checked before the call returns), cfg3, fwd + bwd per view --
    workspaces and gradient tensors allocated per call as torch does for any op.  -> {"ms_per_view", ...}."""
    import numpy as np
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from sings_amd.scene import synthetic_scene
    rank, world, dev, dist, dinfo = ctx
    N, W, H, deg = a.gaussians, a.width, a.height, a.sh_degree
    s = synthetic_scene(N, W, H, deg, 3)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), sh_degree=deg,
                                       campos=t(s["campos"]), prefiltered=False, debug=False)
    req = lambda x: t(x).requires_grad_(True)
    m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
    dL = t(s["dL_dimage"])
    rast = GaussianRasterizer(rs)

    def step(_i=0):
        for x in (m, op, sh, sc, rt):
            x.grad = None
        m2 = torch.zeros_like(m, requires_grad=True)               # gs_renderer_single.py:50-56
        color, radii = rast(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rt)
        color.backward(dL)
    for _ in range(10):
        step()
    n = 100
    els = timed_repeats(None, dev, n, step, min_s=LIGHT_TIMED_S)
    return {"ms_per_view": _median(els) / n * 1e3, "timed_region_s": sum(els), "overflow_check": "sync (default)",
            "workload": f"S({N},{W},{H},deg={deg},seed=3) through diff_gaussian_rasterization.GaussianRasterizer + torch autograd, one view per call"}


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
