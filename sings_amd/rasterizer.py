"""Drop-in replacement of the ``diff_gaussian_rasterization`` Python package on MI355X.

Same public surface the reference imports (sings/rec/renderer/gs_renderer_single.py:6-9,
gs_renderer_multiple.py:6-9; package = install_all.sh:22): ``GaussianRasterizationSettings``
(12-field NamedTuple), ``GaussianRasterizer`` (nn.Module, returns ``(color, radii)``),
``rasterize_gaussians`` and the ``_RasterizeGaussians`` autograd.Function, with the upstream
argument order and gradient order ``(means3D, means2D, sh, colors_precomp, opacities, scales,
rotations, cov3Ds_precomp, None)``.  The compute is the hand-written gfx950 library behind the
C ABI of include/sings_hip.h -- there is no CPU / eager fallback.
"""
import atexit
import ctypes as C
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- pair capacity and the overflow check -------------------------------------------------------------------------
# The workspaces of a call are sized for `cap` (tile, Gaussian) pairs; the forward reports the count R it produced.
# Upstream blocks the host in the MIDDLE of every forward to read R and so never renders with a too small buffer.  Here
# the mode decides when R is read:
#   "sync"     (default; the drop-in guarantee) every forward knows R before it returns and re-runs with a larger
#              workspace if needed: NEVER a wrong frame, whatever the script does between frames.  The count does not wait
#              for the forward to finish: the binning kernel publishes it to a pinned, mapped host word the moment it
#              exists (SgRasterSettings.count_signal, include/sings_hip.h) -- a third of the way into a cfg3 forward -- and
#              the call returns while the composite kernel is still running, so the host queues the loss and the backward
#              behind it instead of starting them after a full stream synchronisation (round 1 / 2's "sync").
#   "async"    opt-in: the forward copies (R, overflow flag) to pinned host memory behind its kernels and returns at once;
#              the value is looked at when the NEXT forward on the device is issued.  The capacity keeps 2x headroom over
#              the largest count seen and the FIRST call of every (device, P, image size) is checked synchronously, so an
#              overflow needs a > 2x jump between two consecutive frames; if it happens, THAT frame rendered the background
#              and got zero gradients (the kernels never follow partly written lists) and the next call raises RuntimeError
#              (``on_overflow="warn"``: a RuntimeWarning instead) after growing the capacity.
#   "deferred" opt-in: no host read at all (a whole step can be captured into a HIP graph): every forward folds (R, flag)
#              into a per-device accumulator ON THE DEVICE; ``check_deferred_overflow()`` reads and resets it.
_capacity_hint = {}                    # device index -> pairs
_mode = {"mode": "sync", "on_overflow": "raise"}
_seen = {}                             # device index -> set of (P, W, H) signatures already checked synchronously
_pending = {}                          # device index -> list of [pinned (R, flag), event, cap] of async forwards
_accum = {}                            # device index -> int32[2] device tensor: max R, OR of flags (deferred mode)
_HEADROOM = 2.0
# "sync" forwards learn, scene by scene, whether SG_FLAG_SHORT_LISTS may be passed: the count word of a frame says whether any tile
# list was longer than 512 entries (SG_COUNT_FLAG_HALF_ROWS); if none was, the next frame of that (device, P, W, H) runs with the
# flag -- no long-list sort launches and, on images of many tiles, direct binning (no scatter pass: include/sings_hip.h).  A frame
# that does meet a list of more than 1024 entries under the flag reports NUM_RENDERED_LONG_LIST and is rendered again without it
# before the call returns: the drop-in guarantee (never a wrong frame) stands.
# Images of few tiles (an avatar frame): the same with SG_COUNT_FLAG_HALF_LONG_ROWS (no list over 8192) and SG_FLAG_LONG_ROWS (rows of
# 16384 keys); the fused LBS ops (sings_amd/skinned.py) pass a learned hint in "deferred" mode / under graph capture too, where a
# violated hint is what a capacity overflow is there: a background frame that check_deferred_overflow() reports.
_short_ok = {}                         # (device index, P, W, H) -> the last frame's lists were all <= 512 entries
_rows_ok = {}                          # (device index, P, W, H) -> ... all <= 8192 entries


def _learn_hints(s, key):
    """After a "sync" forward: read the flags of its count word (if it arrived) into the two hint tables."""
    if not s.count_signal_host:
        return
    word = C.c_uint64.from_address(s.count_signal_host).value
    if word >> 63:                                                   # (not after a timed-out wait: that read the header instead)
        _short_ok[key] = bool((word >> 32) & _lib.COUNT_FLAG_HALF_ROWS)
        _rows_ok[key] = bool((word >> 32) & _lib.COUNT_FLAG_HALF_LONG_ROWS)
_ring = {}                             # device index -> [pinned int32[_RING, 2], next slot]
_RING = 16
_signal = {}                           # device index -> [host address, device address, next slot] of the early-count words
_SIGNAL_SLOTS = 64


_signal_lock = threading.Lock()


def _signal_slot(dev):
    """(device address, host address) of the next early-count word of ``dev`` (a ring: a "sync" forward consumes its word
    before it returns, so slots are only shared by calls that are 64 forwards apart).  Thread-safe (a forward on the main
    thread and one on an autograd / data-loader thread must not be handed the same word); the ring is released at
    interpreter exit only (``reset_overflow_state`` rewinds its cursor: another thread may hold a slot)."""
    with _signal_lock:
        sg = _signal.get(dev.index)
        if sg is None:
            h, d = C.c_void_p(), C.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(_lib.load().sg_signal_alloc(_SIGNAL_SLOTS, C.byref(h), C.byref(d)), "sg_signal_alloc")
            sg = _signal[dev.index] = [h.value, d.value, 0]
        k = sg[2] % _SIGNAL_SLOTS
        sg[2] += 1
        return sg[1] + 8 * k, sg[0] + 8 * k


def _free_signal_rings(dev_index=None):
    with _signal_lock:
        for k in [k for k in _signal if dev_index is None or k == dev_index]:
            sg = _signal.pop(k)
            try:
                _lib.load().sg_signal_free(C.c_void_p(sg[0]))
            except Exception:
                pass


def _free_signal_rings_at_exit():
    # kernels of a forward still in flight write their slot: drain the devices before the pinned words go
    try:
        if torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:
        return                                        # (the process is exiting: leaking 512 pinned bytes is harmless)
    _free_signal_rings()


atexit.register(_free_signal_rings_at_exit)


def set_overflow_check(mode="sync", on_overflow=None, capacity_pairs=None, device=None):
    """Select when the pair count of a forward is checked against the capacity: "sync" (default) | "async" | "deferred"."""
    if mode not in ("async", "sync", "deferred"):
        raise ValueError("mode must be 'async', 'sync' or 'deferred'")
    _mode["mode"] = mode
    if on_overflow is not None:
        if on_overflow not in ("warn", "raise"):
            raise ValueError("on_overflow must be 'warn' or 'raise'")
        _mode["on_overflow"] = on_overflow
    if capacity_pairs is not None:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _capacity_hint[dev.index] = int(capacity_pairs)


def set_deferred_overflow_check(on=True, capacity_pairs=None, device=None):
    """Round-1 name: ``on`` selects "deferred", otherwise back to the default "sync"."""
    set_overflow_check("deferred" if on else "sync", capacity_pairs=capacity_pairs, device=device)


def _grow(dev_index, R):
    _capacity_hint[dev_index] = max(_capacity_hint.get(dev_index, 0), int(R * _HEADROOM) + 1024)


def _report(msg):
    if _mode["on_overflow"] == "raise":
        raise RuntimeError(msg)
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _drain_async(dev, wait_all=False):
    """Look at the (R, flag) copies of earlier async forwards on ``dev``.  Entries older than the newest one are waited
    for (their forward was issued at least one call ago); returns the largest R seen, or None."""
    lst = _pending.get(dev.index)
    if not lst:
        return None
    worst, worst_cap, Rmax = None, None, None
    keep = []
    long_list = False
    for k, (host, ev, cap) in enumerate(lst):
        if not wait_all and k == len(lst) - 1 and not ev.query():
            keep.append(lst[k])                        # the newest one may still be in flight: leave it for the next call
            continue
        ev.synchronize()
        R = int(host[0]) & 0xffffffff
        Rmax = R if Rmax is None else max(Rmax, R)
        long_list = long_list or bool(int(host[1]) & 2)
        if R > cap and (worst is None or R > worst):
            worst, worst_cap = R, cap
    _pending[dev.index] = keep
    if Rmax is not None:
        _grow(dev.index, Rmax)
    if long_list:                                      # a learned list-length hint did not hold for an earlier (fused LBS) forward
        for tbl in (_short_ok, _rows_ok):
            for k in [k for k in tbl if k[0] == dev.index]:
                del tbl[k]
        _report(f"sings_amd: an earlier forward on {dev} met a tile list longer than the row its learned hint promised: that frame "
                f"rendered the background and received zero gradients; the hints of the device are dropped")
    if worst is not None:
        _report(f"sings_amd: an earlier forward on {dev} produced {worst} (tile, Gaussian) pairs for a capacity of {worst_cap}: "
                f"that frame rendered the background and received zero gradients; the capacity is now "
                f"{_capacity_hint[dev.index]} (set_overflow_check('sync') checks every forward before returning)")
    return Rmax


def _after_forward(dev, binning, cap):
    """Called by both autograd functions right after a forward that did not read R synchronously."""
    hdr = binning[:8].view(torch.int32)                # header words 0 (R) and 1 (overflow flag)
    if _mode["mode"] == "deferred" or torch.cuda.is_current_stream_capturing():
        acc = _accum.get(dev.index)
        if acc is None:
            acc = _accum[dev.index] = torch.zeros(2, dtype=torch.int32, device=dev)
        torch.maximum(acc, hdr, out=acc)               # R < 2^31; one tiny launch, capturable into a HIP graph
        return
    ring = _ring.get(dev.index)
    if ring is None:
        ring = _ring[dev.index] = [torch.empty((_RING, 2), dtype=torch.int32).pin_memory(), 0]
    host = ring[0][ring[1] % _RING]                    # (at most two entries are pending at any time: _drain_async)
    ring[1] += 1
    host.copy_(hdr, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _pending.setdefault(dev.index, []).append([host, ev, cap])


def _forward_plan(dev, P, W, H):
    """(capacity, read R synchronously?) for the next forward; drains the async results of earlier calls first."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    mode = _mode["mode"]
    if mode == "async":
        _drain_async(dev)
    cap = max(_capacity_hint.get(dev.index, 0), 4 * P + T, 1 << 16)
    sig = (P, W, H)
    first = sig not in _seen.setdefault(dev.index, set()) or dev.index not in _capacity_hint
    sync = mode == "sync" or (mode == "async" and first)
    if torch.cuda.is_current_stream_capturing():
        sync = False
    return cap, sync, sig


_flag_bit_cache = {}


def _flag_bits(dev):
    """[1, 2, 4, ... 2^15] on ``dev``: OR-reduction of header flag words over the frames of a K-frame call = sum over the bits
    of the per-bit maxima (torch has no bitwise-or reduction)."""
    t = _flag_bit_cache.get(dev.index)
    if t is None:
        t = _flag_bit_cache[dev.index] = (2 ** torch.arange(16, dtype=torch.int32)).to(dev)
    return t


def _forward_done_sync(dev, R, sig):
    _seen[dev.index].add(sig)
    _grow(dev.index, R)


def reset_overflow_state(device=None):
    """Forget capacities, checked signatures and pending results (tests; after a change of scene scale)."""
    for tbl in (_short_ok, _rows_ok):
        for k in [k for k in tbl if device is None or k[0] == torch.device(device).index]:
            del tbl[k]
    for d in (_capacity_hint, _seen, _pending, _accum):
        if device is None:
            d.clear()
        else:
            d.pop(torch.device(device).index, None)
    # The pinned early-count ring is NOT freed here: a forward on another thread may already hold a slot address the
    # scan kernel will write and the host will spin on; only its cursor is rewound.  The ring lives until interpreter exit.
    with _signal_lock:
        for k, sg in _signal.items():
            if device is None or k == torch.device(device).index:
                sg[2] = 0


def check_deferred_overflow(device=None):
    """Largest pair count of the forwards since the last poll on ``device`` (one D2H read = one host synchronisation;
    "deferred" mode: the device-side accumulator; "async" mode: the pending copies).  Raises RuntimeError if one of them
    exceeded its capacity, after growing the capacity used by the following calls.  None if there was no forward."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if _mode["mode"] != "deferred" and dev.index not in _accum:
        old = _mode["on_overflow"]
        _mode["on_overflow"] = "raise"
        try:
            return _drain_async(dev, wait_all=True)
        finally:
            _mode["on_overflow"] = old
    acc = _accum.get(dev.index)
    if acc is None:
        return None
    R, flag = (int(v) for v in acc.cpu())              # largest R, OR of the overflow flags (each set on the device against
    acc.zero_()                                        # the capacity THAT forward ran with) since the last poll
    if R == 0 and flag == 0:
        return None
    _grow(dev.index, R)
    if flag & 2:                                       # a learned list-length hint was violated: forget the hints of this device
        for tbl in (_short_ok, _rows_ok):
            for k in [k for k in tbl if k[0] == dev.index]:
                del tbl[k]
    if flag:
        raise RuntimeError(f"sings_amd: a deferred forward produced up to {R} (tile, Gaussian) pairs, more than its capacity"
                           f"{' (or met a tile list longer than the row its learned hint promised: the hints are dropped)' if flag & 2 else ''}: "
                           f"it rendered the background and no gradients; the capacity is now {_capacity_hint[dev.index]}")
    return R


def _ptr(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _f32(t, name, dev):
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a tensor")
    if t.device != dev:
        raise RuntimeError(f"{name} is on {t.device}, expected {dev}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


def _settings_struct(rs, dev, sh_coeffs, keep):
    bg = _f32(rs.bg, "bg", dev); vm = _f32(rs.viewmatrix, "viewmatrix", dev)
    pm = _f32(rs.projmatrix, "projmatrix", dev); cp = _f32(rs.campos, "campos", dev)
    keep.extend([bg, vm, pm, cp])
    s = _lib.SgRasterSettings()
    s.image_height = int(rs.image_height); s.image_width = int(rs.image_width)
    s.tanfovx = float(rs.tanfovx); s.tanfovy = float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree); s.sh_coeffs = int(sh_coeffs)
    s.prefiltered = int(bool(rs.prefiltered)); s.debug = int(bool(rs.debug)); s.flags = 0
    s.bg = bg.data_ptr(); s.viewmatrix = vm.data_ptr(); s.projmatrix = pm.data_ptr(); s.campos = cp.data_ptr()
    s.count_signal = None; s.count_signal_host = None
    return s


def _arm_early_count(s, dev):
    """Give the settings of a forward that wants R back an early-count word (see "sync" above)."""
    s.count_signal, s.count_signal_host = _signal_slot(dev)


def _backward_buffers(dev, P, M, bwd_bytes, has_sh, has_col, has_scale, has_cov):
    """The gradient tensors and the record workspace of one backward call (every element is written by the kernels)."""
    def e(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)
    return (e(P, 3), e(P, 3), e(P, 1), e(P, M, 3) if has_sh else None, e(P, 3) if has_col else None,
            e(P, 3) if has_scale else None, e(P, 4) if has_scale else None, e(P, 6) if has_cov else None,
            torch.empty(int(bwd_bytes), dtype=torch.uint8, device=dev))


_PREALLOC_SETS = 2                   # backward-buffer sets a forward may allocate ahead of their backward (two views in flight)
_PREALLOC = {"out": 0}


class _PreallocSet:
    """A forward's pre-allocated backward buffers; counted while they wait (a graph dropped without a backward frees the count too)."""

    def __init__(self, bufs):
        self.bufs = bufs
        _PREALLOC["out"] += 1

    def take(self):
        bufs, self.bufs = self.bufs, None
        if bufs is not None:
            _PREALLOC["out"] -= 1
        return bufs

    def __del__(self):
        self.take()


def _empty_to_none(t):
    return None if t is None or t.numel() == 0 else t


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, write_point_keys=False):
        lib = _lib.load()
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError("sings_amd rasterizer runs on the MI355X only (tensors must be on a 'cuda' "
                               "device); there is no CPU fallback")
        rs = raster_settings
        means3D = _f32(means3D, "means3D", dev)
        sh = _f32(_empty_to_none(sh), "sh", dev)
        colors_precomp = _f32(_empty_to_none(colors_precomp), "colors_precomp", dev)
        opacities = _f32(opacities, "opacities", dev)
        scales = _f32(_empty_to_none(scales), "scales", dev)
        rotations = _f32(_empty_to_none(rotations), "rotations", dev)
        cov3Ds_precomp = _f32(_empty_to_none(cov3Ds_precomp), "cov3Ds_precomp", dev)
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        M = int(sh.shape[1]) if sh is not None else 0
        keep = []
        s = _settings_struct(rs, dev, M, keep)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cap, sync, sig = _forward_plan(dev, P, W, H)
        need_bwd = any(ctx.needs_input_grad[:8])
        hint_key = (dev.index,) + sig
        hint = sync and not rs.debug and _short_ok.get(hint_key, False)
        rows = sync and not rs.debug and _rows_ok.get(hint_key, False)       # (images of few tiles; the other flag is ignored there)
        with torch.cuda.device(dev):
            if sync and not rs.debug:
                _arm_early_count(s, dev)
            while True:
                L = _lib.layout(P, W, H, cap)
                geom = torch.empty(L.geom_bytes, dtype=torch.uint8, device=dev)
                binning = torch.empty(L.bin_bytes, dtype=torch.uint8, device=dev)
                img = torch.empty(L.img_bytes, dtype=torch.uint8, device=dev)
                # The backward's buffers are allocated HERE, in front of the forward's launch + wait for the pair count: the host
                # work of a "sync" forward ends when the scan kernel has published R, and from there to the backward's first
                # launch the GPU has only the forward composite (~70 us at cfg3) to run -- autograd's thread hop plus eight
                # allocations, a layout call and a settings struct on the far side of that wait left it idle for ~50 us per view
                # (drop-in surface 0.358 ms against 0.293 for the pre-allocated engine; now the allocations hide under the previous
                # view's backward).  Only when a gradient is wanted; handed out once (a second backward through a retained graph
                # allocates afresh: autograd may have adopted the first set as .grad).
                # (At most _PREALLOC_SETS such sets are outstanding: a caller that renders many views before ONE backward would hold a
                #  set -- ~80 MB at cfg3 -- per view; its later forwards leave the allocation to the backward, as before round 5.)
                bufs = None
                if need_bwd and _PREALLOC["out"] < _PREALLOC_SETS:
                    bufs = _backward_buffers(dev, P, M, L.bwd_bytes, sh is not None, colors_precomp is not None, scales is not None,
                                             cov3Ds_precomp is not None)
                nr = C.c_int64(0)
                s.flags = (_lib.FLAG_SHORT_LISTS if hint else 0) | (_lib.FLAG_LONG_ROWS if rows else 0)
                _lib.check(lib.sg_rasterize_forward(
                    C.byref(s), P, _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
                    _ptr(rotations), _ptr(cov3Ds_precomp), _ptr(geom), _ptr(binning), cap, _ptr(img),
                    _ptr(color), _ptr(radii), int(bool(write_point_keys)), C.byref(nr) if sync else None, stream), "forward")
                R = int(nr.value) if sync else None
                if (hint or rows) and R == _lib.NUM_RENDERED_LONG_LIST:       # the scene changed under the hint: this frame again, without it
                    _short_ok[hint_key] = hint = False
                    _rows_ok[hint_key] = rows = False
                    continue
                if not sync or R <= cap:
                    break
                cap = int(R * _HEADROOM) + 1024     # workspace too small: grow and re-run
            if sync:
                _forward_done_sync(dev, R, sig)
                _learn_hints(s, hint_key)
            else:
                _after_forward(dev, binning, cap)
        ctx.raster_settings = rs
        ctx.num_rendered = R
        ctx.capacity = cap
        ctx.sh_coeffs = M
        ctx.bwd_bufs = None if bufs is None else _PreallocSet(bufs)
        ctx.bwd_bytes = int(L.bwd_bytes)
        ctx.settings_struct = (s, keep)                     # the backward passes the same struct (same camera, same flags)
        ctx.flags = (sh is not None, colors_precomp is not None, scales is not None, cov3Ds_precomp is not None)
        z = torch.empty(0, device=dev)
        ctx.save_for_backward(means3D, sh if sh is not None else z,
                              colors_precomp if colors_precomp is not None else z,
                              opacities, scales if scales is not None else z,
                              rotations if rotations is not None else z,
                              cov3Ds_precomp if cov3Ds_precomp is not None else z,
                              radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii=None):
        lib = _lib.load()
        (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, radii, geom, binning,
         img) = ctx.saved_tensors
        has_sh, has_col, has_scale, has_cov = ctx.flags
        rs = ctx.raster_settings
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        s, keep = ctx.settings_struct
        s.count_signal = None; s.count_signal_host = None
        g = _f32(grad_out_color, "grad_out_color", dev)
        held, ctx.bwd_bufs = ctx.bwd_bufs, None
        bufs = None if held is None else held.take()
        if bufs is None:
            bufs = _backward_buffers(dev, P, ctx.sh_coeffs, ctx.bwd_bytes, has_sh, has_col, has_scale, has_cov)
        dmeans3D, dmeans2D, dopac, dsh, dcol, dscales, drots, dcov, bwd_ws = bufs
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_rasterize_backward(
                C.byref(s), P, _ptr(means3D), _ptr(sh) if has_sh else None,
                _ptr(colors_precomp) if has_col else None, _ptr(opacities),
                _ptr(scales) if has_scale else None, _ptr(rotations) if has_scale else None,
                _ptr(cov3Ds_precomp) if has_cov else None, _ptr(radii), _ptr(geom), _ptr(binning), ctx.capacity,
                _ptr(img), _ptr(bwd_ws), _ptr(g), _ptr(dmeans3D), _ptr(dmeans2D), _ptr(dsh), _ptr(dcol),
                _ptr(dopac), _ptr(dscales), _ptr(drots), _ptr(dcov), stream), "backward")
        return (dmeans3D, dmeans2D, dsh, dcol, dopac.view_as(opacities), dscales, drots, dcov, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        lib = _lib.load()
        rs = self.raster_settings
        with torch.no_grad():
            dev = positions.device
            pos = _f32(positions, "positions", dev)
            vm = _f32(rs.viewmatrix, "viewmatrix", dev)
            pm = _f32(rs.projmatrix, "projmatrix", dev)
            out = torch.empty(pos.shape[0], dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.sg_mark_visible(int(pos.shape[0]), _ptr(pos), _ptr(vm), _ptr(pm), _ptr(out),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           "mark_visible")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D '
                            'covariance!')
        empty = torch.Tensor([])
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
