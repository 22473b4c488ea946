"""bench.py --workload raster (the headline: cfg3; cfg2 / cfg5 by flags) and the drop-in autograd leg."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

from .distrib import (COMM_KEYS, FORCE_DIST, LIGHT_TIMED_S, MAX_REPEATS, MIN_TIMED_S, _grad_sha256, _log, _median, _ranks_agree,   # noqa: F401
                      allreduce_probe, densification_stats_check, dp_self_check, exposed_by_algorithm, make_frame_parallel,
                      one_view_step_by_algorithm, timed_region, timed_repeats, usable_cores)
from .roofline import (HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, algorithmic_bytes, algorithmic_bytes_skinned, build_roofline, measure_copy_peak,   # noqa: F401
                       pmc_view_traffic, scaling_model, train_step_roofline)
from .baseline import PARITY_BORDER, PARITY_RGB_TOL, cpu_baseline    # noqa: F401
from .common import _release, _tile_list_stats


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
        # violation surfaces in num_rendered() below).  Either flag selects direct binning (many tiles: SG_FLAG_SHORT_LISTS, few tiles, e.g.
        # cfg2's 1 024: SG_FLAG_LONG_ROWS; each is ignored in the other regime)
        if Kf == 1:
            e = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=int(R * 1.1) + 4096, grad_flat=grads[v % rows])
            e.set_camera(camera(rank * k_views + v)[3], short_lists=short, long_rows=short)
        else:
            from sings_amd.engine import RasterFramesEngine
            e = RasterFramesEngine(N, W, H, shs.shape[1], Kf, dev, capacity_pairs=int(R * 1.1) + 4096, grad_flat=grads[v % rows])
            cams = [camera(rank * k_views + v * Kf + f) for f in range(Kf)]
            e.set_camera(cams[0][3]._replace(viewmatrix=t(np.stack([c_[0] for c_ in cams])), projmatrix=t(np.stack([c_[1] for c_ in cams])),
                                             campos=t(np.stack([c_[2] for c_ in cams]))), short_lists=short, long_rows=short)
        engs.append(e)
    if Kf == 1:
        eng = engs[0]
    else:                                                        # the one-view-per-step leg and the parity views: a plain engine
        eng = RasterEngine(N, W, H, shs.shape[1], dev, capacity_pairs=int(R * 1.1) + 4096)
        eng.set_camera(camera(rank * k_views)[3], short_lists=short, long_rows=short)
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
