"""bench.py --workload avatar (cfg4): LBS-fused forward + photometric loss + backward over posed frames, K frames per launch."""
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
from .baseline import PARITY_BORDER, PARITY_RGB_TOL, _ParityLog, cpu_lbs_project    # noqa: F401
from .common import _release, _tile_list_stats


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
    Rmax, tile_mean, tile_max, longest = 0, 0.0, 0, 0
    for f in range(0, F, 8):
        eng.set_frame(xyz, None, w, A_all[f], smpl_scale, transl)
        Rf = eng.forward(sh, op, sc, sync_num_rendered=True)
        tm_, tx_ = _tile_list_stats(eng, W, H)
        longest = max(longest, tx_)
        if Rf > Rmax:
            Rmax = Rf
            tile_mean, tile_max = tm_, tx_
    # the sizing pass knows the longest tile list of the sequence (~1e4): with 1.3x margin it fits a 16384-key row, so the frames are
    # binned directly (SG_FLAG_LONG_ROWS; checked on the device, a violation surfaces as NUM_RENDERED_LONG_LIST in num_rendered())
    long_rows = longest * 1.3 <= 16384
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
        e.set_camera(rs, long_rows=long_rows)
        engs.append(e)
        losses.append(PhotoLossEngine(W, H, dev, l1_w=0.8, ssim_w=0.2, K=Kf))    # human.loss.l1_w / ssim_w
    eng = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=int(Rmax * 1.3) + 4096, sh_planar=True) if Kf > 1 else engs[0]
    active = eng.active_floats(0)                                    # N * 10 of the N * 55 floats
    if Kf > 1:
        eng.set_camera(rs, long_rows=long_rows)
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
    assert all((0 <= min(e.num_rendered()) and max(e.num_rendered()) <= e.cap) if Kf > 1 else 0 <= e.num_rendered() <= e.cap for e in engs), \
        "pair capacity / long-rows hint violated"
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
    eng.set_camera(rs, long_rows=bool(getattr(eng, "_rows", 0)))
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
