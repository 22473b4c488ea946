"""bench.py --workload train: the complete training step (decode -> LBS-fused raster -> L1 / SSIM -> regularisers -> backward) from one HIP graph."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

from .distrib import (COMM_KEYS, FORCE_DIST, LIGHT_TIMED_S, MAX_REPEATS, MIN_TIMED_S, _grad_sha256, _log, _median, _ranks_agree,   # noqa: F401
                      allreduce_probe, densification_stats_check, dp_self_check, exposed_by_algorithm, make_frame_parallel,
                      one_view_step_by_algorithm, timed_region, timed_repeats, usable_cores)
from .baseline import PARITY_BORDER, PARITY_RGB_TOL
from .roofline import (HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, algorithmic_bytes, algorithmic_bytes_skinned, build_roofline, measure_copy_peak,   # noqa: F401
                       pmc_view_traffic, scaling_model, train_step_roofline)


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
