"""Rooflines of the bench line: SURVEY.md 8(d) algorithmic bytes, the measured copy peak, the committed PMC passes with their sidecars
(a pass is used only if it profiled THIS configuration and THIS tree), the 8-GPU scaling MODEL.  No oracle, no workload."""
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0
XGMI_LINK_GBS = 153.0          # per direction and link; 7 links per GPU
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs
VALU_CYCLES_GUIDE = 2.0        # MI355X_MICROARCH.md constants table: one wave64 v_fma_f32 issues in 2 cycles
VALU_CYCLES_MIX = 3.6          # issue cost of the backward composite's instruction mix, from tools/valu_probe.hip's per-kind costs: per
                               # pass 6 lane swaps x 8.2 + 13 DPP x 4.2 + exp2, rcp x 8.2 + 6 cmp / select x 3.9 + ~45 plain x 2.6 cycles over 72
                               # instructions (round 1's 95-instruction pass: 2.9 -- the instructions removed since were the cheap ones)


def algorithmic_bytes(N, H, W, R, deg):
    """SURVEY.md 8(d) per-unit figures, split per kernel (DESIGN.md sections 4-5)."""
    inb = 44 + 12 * (deg + 1) ** 2
    rec, gout, hw = 75, 248, H * W
    per = {
        "sg_preprocess_fwd_kernel": N * (inb + 4 + rec),
        "binning": R * 12,
        "sg_render_fwd_kernel": N * rec + hw * (12 + 8) + R * 4,
        "sg_render_bwd_kernel": hw * (12 + 8) + R * 4,
        "sg_preprocess_bwd_kernel": N * (inb + gout),
    }
    total = N * (2 * inb + gout + 4 + 2 * rec) + hw * 40 + R * 20
    return per, total


def algorithmic_bytes_skinned(N, H, W, R, deg, J, has_rot=False):
    """The same for the LBS-fused path (last sentences): + N (12 + [36] + 4 J) read forward and again backward
    (xyz_canon, [R_canon], skinning weights), + N (12 [+ 36]) written backward (dL/dxyz_canon [, dL/dR_canon]); posed means /
    quaternions are never materialised: - N 28 read per pass, - N 28 of gradient writes."""
    per, total = algorithmic_bytes(N, H, W, R, deg)
    extra_in = N * (12 + (36 if has_rot else 0) + 4 * J) - N * 28
    extra_out = N * (12 + (36 if has_rot else 0)) - N * 28
    per = dict(per)
    per["sg_preprocess_fwd_kernel"] += extra_in
    per["sg_preprocess_bwd_kernel"] += extra_in + extra_out
    return per, total + 2 * extra_in + extra_out


MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA (v_mfma_f32_32x32x2_f32), dense = the fp32 vector peak


def train_step_roofline(N, K, modules, W, H, R, J, ms_per_step, copy_gbs):
    """The complete training step (decode -> LBS-fused raster -> loss -> backward) against its two roofs: the decoders' linear layers on the
    fp32 matrix cores and the bytes every stage has to move once.  `modules` = (tri-plane, geometry decoder, appearance decoder).
      FLOPs  every nn.Linear of the two decoders: 2 N Cin Cout forward, the same again for the input gradient and for the weight gradient
             (the first layers' input gradient included: the tri-plane features carry gradient to the planes)
      bytes  per linear layer N 4 (3 Cin + 2 Cout + 2 Cout [activation]): x read forward and again for the weight gradient, h written,
             dh read, dx written, the activation's saved value written and read;  tri-plane: the planes read forward, their gradient
             written (2 x plane bytes; the 12 x 4 texel gathers per point hit L2 / MALL), xyz in, features out, their gradient in;
             raster + LBS: algorithmic_bytes_skinned per frame; loss: 40 B per pixel and frame;  k-NN and L2 regularisers: N (12 + 4 + 4)
             read, N 4 written (their neighbour search is compute, not traffic).
    -> the roofline block: bound = whichever roof takes longer at its peak; frac = that time / the measured step."""
    import torch
    tri, geo, app = modules
    lin = [m for mod in (geo, app) for m in mod.modules() if isinstance(m, torch.nn.Linear)]
    acts = {id(m) for mod in (geo, app) for seq in mod.modules() if isinstance(seq, torch.nn.Sequential)
            for a_, b_ in zip(list(seq), list(seq)[1:]) if isinstance(a_, torch.nn.Linear) and not isinstance(b_, torch.nn.Linear) for m in (a_,)}
    flops = sum(3 * 2 * N * m.in_features * m.out_features for m in lin)
    b_lin = sum(N * 4 * (3 * m.in_features + 2 * m.out_features + (2 * m.out_features if id(m) in acts else 0)) for m in lin)
    plane_bytes = sum(p.numel() * 4 for p in tri.parameters())
    feat = lin[0].in_features
    b_tri = 2 * plane_bytes + N * (12 + 2 * 4 * feat + 12)
    _, b_raster = algorithmic_bytes_skinned(N, H, W, R, 0, J, has_rot=False)
    b_loss = 40 * W * H
    b_reg = N * (12 + 4 + 4 + 4)
    total = b_lin + b_tri + K * (b_raster + b_loss) + b_reg
    t_mfma = flops / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3
    t_hbm = total / (copy_gbs * 1e9) * 1e3
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    return {"bound": bound, "scope": "whole_step", "frac": max(t_mfma, t_hbm) / ms_per_step,
            "achieved": flops / (ms_per_step * 1e-3) / 1e12 if bound == "mfma" else total / (ms_per_step * 1e-3) / 1e9,
            "peak": MFMA_F32_PEAK_TFLOPS if bound == "mfma" else copy_gbs, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "mfma": {"flops_per_step": flops, "ms_at_peak": t_mfma, "frac": t_mfma / ms_per_step, "peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                     "achieved_TFLOPs": flops / (ms_per_step * 1e-3) / 1e12, "note": "fp32-input MFMA, dense; the decoders' nn.Linear layers only"},
            "hbm": {"algorithmic_bytes_per_step": total, "ms_at_copy_rate": t_hbm, "frac": t_hbm / ms_per_step, "peak_GBs": copy_gbs,
                    "achieved_GBs": total / (ms_per_step * 1e-3) / 1e9,
                    "split": {"linear_layers": b_lin, "triplane": b_tri, "raster_lbs": K * b_raster, "loss": K * b_loss, "regularisers": b_reg}},
            "traffic": None,
            "note": "both roofs are lower bounds that overlap at best: the step cannot be shorter than the larger one; roofs in sequence "
                    "(ms_at_peak + ms_at_copy_rate) are the bound of a step that overlaps nothing"}


def measure_copy_peak(dev, gib=1.0):
    """float4-copy bandwidth of THIS box in GB/s: sg_copy_probe (a plain 16-byte-per-lane copy kernel) over `gib` GiB, HIP events
    on the launch stream, best of 3 after one warm-up; bytes = read + written.  SURVEY.md 8(d) / BASELINE.md section 2: the
    denominator of the HBM roofline is measured, not quoted (the guide's figure for this part is 6.29 TB/s)."""
    import torch
    from sings_amd import _lib
    lib = _lib.load()
    n = int(gib * (1 << 30)) & ~255
    src = torch.zeros(n, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    best = 0.0
    for nt in (0, 1):                                  # plain / non-temporal loads + stores: the better form is the box's copy rate
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.sg_copy_probe(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, nt, st), "copy probe")
            e1.record()
            torch.cuda.synchronize(dev)
            if i:
                best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    torch.cuda.empty_cache()
    return best


def scaling_model(bytes_, t_step_ms, t_one_ms, exposed_ms=None, world=8):
    """A clearly labelled MODEL of the 8-GPU frame-parallel step (no multi-GPU box is available to this build: the driver's SCALE
    run is the measurement this is to be compared with).  Per step every rank adds ONE collective over `bytes_` of gradient:
      rs_ag      : reduce-scatter + all-gather, every rank exchanging 1/W of the buffer with each peer over its own xGMI link:
                   2 (W-1)/W S / ((W-1) 153 GB/s) = 2 S / (W 153 GB/s)
      all_reduce : one ring: 2 (W-1)/W S / 153 GB/s  (per-link bound of RCCL's default schedule on a point-to-point mesh)
    and cannot hide it (every gradient element depends on the last backward kernel; the optimiser needs the sum), so
      predicted_scale_W = W t_step / (t_step + max(exposed_measured_world1, t_collective))
    at the batched step and at the reference's one frame per step."""
    S = float(bytes_)
    link = XGMI_LINK_GBS * 1e9
    coll = {"rs_ag": 2.0 * S / (world * link) * 1e3, "all_reduce": 2.0 * (world - 1) / world * S / link * 1e3}
    floor = exposed_ms or 0.0
    pred = {}
    for label, t in (("batched", t_step_ms), ("one_view_per_step", t_one_ms)):
        pred[label] = {k: world * t / (t + max(floor, v)) for k, v in coll.items()}
    return {"label": "MODEL of the 8-GPU step, not a measurement (SCALE runs are the driver's)", "world": world, "collective_bytes": int(S),
            "xgmi_link_GBs": XGMI_LINK_GBS, "collective_ms": coll, "exposed_ms_measured_world1": exposed_ms,
            "t_step_ms": {"batched": t_step_ms, "one_view_per_step": t_one_ms}, "predicted_scale_8": pred}
# kernel sources a PMC pass is keyed to (git blob hashes in the sidecar): the rasterisation path ...
RASTER_SOURCES = ("sings_amd/csrc/sg_render.hip", "sings_amd/csrc/sg_sort.h", "sings_amd/csrc/sg_binning.hip",
                  "sings_amd/csrc/sg_project.h", "sings_amd/csrc/sg_preprocess.hip", "sings_amd/csrc/sg_skin.hip",
                  "sings_amd/csrc/sg_common.h", "sings_amd/csrc/sg_math.h")
# ... and (round 6) the loss / decode / linear / regulariser / rotation kernels of the training step
PMC_SOURCES = RASTER_SOURCES + ("sings_amd/csrc/sg_loss.hip", "sings_amd/csrc/sg_decode.hip", "sings_amd/csrc/sg_linear.hip",
                                "sings_amd/csrc/sg_reg.hip", "sings_amd/csrc/sg_rot.hip", "sings_amd/csrc/sg_rot.h")
# which of them a kernel is compiled from (by name prefix; everything else: the rasterisation path)
KERNEL_SOURCES = (
    (("sg_photo", "sg_mask_sum", "sg_loss_reduce", "sg_ssim"), ("sings_amd/csrc/sg_loss.hip", "sings_amd/csrc/sg_common.h")),
    (("sg_triplane", "sg_tp_", "sg_bias_act"), ("sings_amd/csrc/sg_decode.hip", "sings_amd/csrc/sg_common.h")),
    (("sg_linear", "sg_weight_grad", "sg_wgrad"), ("sings_amd/csrc/sg_linear.hip", "sings_amd/csrc/sg_common.h")),
    (("sg_region", "sg_rows_lap", "sg_mesh_edge", "sg_l2norm", "sg_knn", "sg_reg"), ("sings_amd/csrc/sg_reg.hip", "sings_amd/csrc/sg_common.h")),
    (("sg_m2q", "sg_rot", "sg_qmul", "sg_joint"), ("sings_amd/csrc/sg_rot.hip", "sings_amd/csrc/sg_rot.h", "sings_amd/csrc/sg_common.h")),
)


def sources_of(kernel=None, workload="raster"):
    """The source files whose hashes decide whether a committed PMC pass still describes `kernel` (None: every kernel the
    workload launches -- the rasterisation path for 'raster', all of PMC_SOURCES for 'avatar' / 'train')."""
    if kernel is not None:
        base = kernel.split("<")[0]
        for prefixes, files in KERNEL_SOURCES:
            if base.startswith(prefixes):
                return files
        return RASTER_SOURCES
    return RASTER_SOURCES if workload == "raster" else PMC_SOURCES


def git_blob_sha1(path):
    """The hash `git hash-object` gives the file (no git needed on the GPU box)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes(root=ROOT):
    return {rel: git_blob_sha1(os.path.join(root, rel)) for rel in PMC_SOURCES}


def _meta_status(meta, cfg, root=ROOT, kernel=None):
    """None if the PMC pass described by `meta` (sidecar written by tools/pmc_summary.py: configuration it ran + git blob
    hashes of the kernel sources it profiled) is a pass of THIS configuration over THIS tree; otherwise the reason.  The sources
    that count are those `kernel` is compiled from (sources_of; None: all the workload's kernels): a sidecar that does not key one
    of them (passes older than round 6 key the rasterisation path only) says nothing about that kernel."""
    if not isinstance(meta, dict) or "sources" not in meta or "config" not in meta:
        return "no sidecar (configuration and kernel-source hashes of the PMC pass unknown)"
    if any(meta["config"].get(k) != v for k, v in cfg.items()):
        return f"PMC pass of another configuration ({meta['config']})"
    cur = source_hashes(root)
    need = sources_of(kernel, cfg.get("workload", "raster"))
    missing = sorted(rel for rel in need if rel not in meta["sources"])
    if missing:
        return "the PMC pass is not keyed to " + ", ".join(os.path.basename(c) for c in missing)
    changed = sorted(rel for rel in need if cur.get(rel) != meta["sources"][rel])
    if changed:
        return "kernel sources changed since the PMC pass: " + ", ".join(os.path.basename(c) for c in changed)
    return None


# kernels behind one event bracket of the library's profiler (sg_profile_collect): few-tile frames take the deep forward and
# zero the records + run the sparse backward inside the "sg_render_bwd_kernel" bracket
KERNEL_VARIANTS = {"sg_render_fwd_kernel": ("sg_render_fwd_kernel", "sg_render_fwd_deep_kernel"),
                   "sg_render_bwd_kernel": ("sg_render_bwd_kernel", "sg_render_bwd_sparse_kernel", "sg_order_items_kernel")}


def _committed_pmc(kernel, cfg, pdir=None, root=ROOT):
    """VALU wave-instructions and HBM bytes per launch of `kernel` from the newest committed PMC passes of THIS configuration
    (profiles/<tag>_pmc_SQ.csv + <tag>_pmc_SQ.meta.json, profiles/hbm_traffic.json with its "_meta"; tools/pmc_summary.py writes
    them).  A pass whose sidecar names another configuration, or whose recorded source hashes differ from the tree, is NOT
    used: the counts describe other kernels -- the caller then falls back to the HBM roofline and says why (`stale`)."""
    import csv
    res = {"stale": None}
    pdir = os.path.join(ROOT, "profiles") if pdir is None else pdir
    why = "no committed *_pmc_SQ.csv"
    try:
        for fn in sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_SQ.csv")):
            try:
                meta = json.load(open(os.path.join(pdir, fn[:-4] + ".meta.json")))
            except Exception:
                meta = None
            st = _meta_status(meta, cfg, root, kernel)
            if st is not None:
                why = f"profiles/{fn}: {st}"
                continue
            names = KERNEL_VARIANTS.get(kernel, (kernel,))
            got = [float(r["mean"]) for r in csv.DictReader(open(os.path.join(pdir, fn)))
                   if r["kernel"].split("<")[0] in names and r["Counter_Name"] == "SQ_INSTS_VALU"]
            if got:
                res["valu"], res["valu_source"] = sum(got), f"profiles/{fn}"
            else:
                why = f"profiles/{fn}: no SQ_INSTS_VALU row for {kernel}"
    except Exception as e:
        why = f"{type(e).__name__}: {e}"
    if "valu" not in res:
        res["stale"] = why
    try:
        for fn in sorted(f for f in os.listdir(pdir) if f.endswith("hbm_traffic.json")):
            tj = json.load(open(os.path.join(pdir, fn)))
            st = _meta_status(tj.get("_meta"), cfg, root, kernel)
            if st is None:
                got = [v for k, v in tj.items() if k.split("<")[0] in KERNEL_VARIANTS.get(kernel, (kernel,))]
                res["traffic"] = sum(got) if got else None
                res["traffic_source"] = f"profiles/{fn}"
                res.pop("traffic_stale", None)
                break
            res["traffic_stale"] = f"{fn}: {st}"
    except Exception:
        pass
    return res


def pmc_view_traffic(cfg, pdir=None, root=ROOT, frames=1):
    """Sum over ALL kernels of the committed HBM-traffic pass of this configuration and tree (bytes per VIEW), or None.  A pass
    is taken with ONE view per launch or with K frames / cameras per launch (sidecar key `frames_per_launch`; the per-launch
    counts are divided by K by tools/pmc_summary.py): the pass of the run's own K is preferred, the one-view pass is the fallback.
    -> (bytes, source, frames per launch of the pass)."""
    pdir = os.path.join(ROOT, "profiles") if pdir is None else pdir
    best = None
    try:
        for fn in sorted((f for f in os.listdir(pdir) if f.endswith("hbm_traffic.json")), reverse=True):
            tj = json.load(open(os.path.join(pdir, fn)))
            meta = tj.get("_meta")
            if _meta_status(meta, cfg, root) is None:
                k = int(meta["config"].get("frames_per_launch", 1))
                # (the raster line's one-view TRAIN step also runs the photometric-loss kernels: they are not part of a rendered
                #  view -- neither of `value` nor of its algorithmic bytes -- and stay out of its traffic)
                skip = ("sg_ssim", "sg_loss", "sg_photo", "sg_mask_sum") if cfg.get("workload") == "raster" else ()
                tot = sum(v for kk, v in tj.items() if not kk.startswith("_") and isinstance(v, (int, float)) and not kk.startswith(skip))
                if k == frames:
                    return tot, f"profiles/{fn}", k
                if k == 1 and best is None:
                    best = (tot, f"profiles/{fn}", 1)
    except Exception:
        pass
    return best if best else (None, None, None)


def build_roofline(kern, per, cfg, total_bytes, s_per_view, copy_gbs, frames=1):
    """-> (roofline, roofline_valu).

    roofline (SURVEY.md 8(d) "Which roofline" / "Algorithmic bytes per view"): bound "hbm", scope the WHOLE pass of one view --
    `achieved` = algorithmic bytes per view / seconds per view of the timed region, `peak` = the float4-copy bandwidth measured in
    this run (`peak_spec` = the 8 TB/s of the data sheet, `frac_of_spec` against it), `traffic` = HBM bytes per view summed over
    the kernels of the committed PMC pass of this configuration and tree (null if none matches).  The dominant kernel rides
    along: its own algorithmic bytes / its live HIP-event duration (`dominant_kernel_frac`, same peak).
    roofline_valu: the secondary bound SURVEY.md 8(d) names -- VALU issue of the dominant composite kernel (instructions per
    launch from the committed PMC pass x the guide's 2 cycles / (duration x 2.4 GHz x 1024 SIMDs)); null without a matching pass."""
    dom = max((k for k in per if k in kern), key=lambda k: kern[k])
    pmc = _committed_pmc(dom, cfg)
    peak = copy_gbs if copy_gbs else HBM_COPY_GBS
    ach = total_bytes / s_per_view / 1e9
    dom_ach = per[dom] / (kern[dom] * 1e-3) / 1e9
    view_traffic, tsrc, tframes = pmc_view_traffic(cfg, frames=frames)
    roof = {"bound": "hbm", "scope": "whole_pass", "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
            "peak_source": "float4 copy measured in this run (sg_copy_probe, 1 GiB, best of 3, plain or non-temporal)" if copy_gbs else
                           "MI355X_MICROARCH.md (6.29 TB/s float4 copy; not measured in this run)",
            "peak_spec": HBM_PEAK_GBS, "frac_of_spec": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_view": total_bytes, "ms_per_view": s_per_view * 1e3,
            "traffic": view_traffic, "traffic_source": tsrc, "traffic_frames_per_launch": tframes,
            "dominant_kernel": dom, "dominant_kernel_ms": kern[dom], "dominant_kernel_algorithmic_bytes": per[dom],
            "dominant_kernel_achieved": dom_ach, "dominant_kernel_frac": dom_ach / peak,
            "dominant_kernel_frac_of_spec": dom_ach / HBM_PEAK_GBS, "dominant_kernel_traffic": pmc.get("traffic")}
    if pmc.get("traffic_stale"):
        roof["traffic_note"] = "profiles/hbm_traffic.json not used: " + pmc["traffic_stale"]
    if dom not in ("sg_render_bwd_kernel", "sg_render_fwd_kernel"):
        return roof, None
    if not pmc.get("valu"):
        return roof, {"bound": "valu", "kernel": dom, "frac": None,
                      "note": "the composite kernels are VALU-issue bound (DESIGN.md section 4); omitted because no PMC pass "
                              "matches this run -- " + str(pmc["stale"])}
    instr = pmc["valu"]
    rate = instr / (kern[dom] * 1e-3) / 1e9                                     # G wave-instructions / s
    vpeak = SIMDS * CLOCK_HZ / VALU_CYCLES_GUIDE / 1e9
    cpi = kern[dom] * 1e-3 * CLOCK_HZ * SIMDS / instr
    return roof, {"bound": "valu", "kernel": dom, "achieved": rate, "peak": vpeak, "unit": "G wave64-instr/s",
                  "frac": rate / vpeak, "traffic": pmc.get("traffic"), "kernel_ms": kern[dom],
                  "valu_wave_instructions_per_launch": instr, "source": pmc.get("valu_source"),
                  "cycles_per_instruction": cpi, "peak_cycles_per_instruction": VALU_CYCLES_GUIDE,
                  "frac_vs_measured_mix_cost": VALU_CYCLES_MIX / cpi,
                  "note": "secondary bound (SURVEY.md 8(d)): peak = guide's 2 cycles per wave64 VALU instruction; "
                          "frac_vs_measured_mix_cost uses the cycles per instruction that tools/valu_probe.hip's per-kind costs "
                          "give for this kernel's instruction mix"}
