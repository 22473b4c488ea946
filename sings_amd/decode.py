"""Attribute decode upstream of the render path: tri-plane features + the two decoder MLPs.

Module and parameter names mirror the reference so that its checkpoints load unchanged
(``SinGS.state_dict()['triplane']``, ``['geometry_dec_i']``, ``['appearance_dec_i']``):

    HexPlaneField(planeconfig, bounds)        sings/rec/models/modules/hexplane.py:107-190   (grids.{s}.{c}, aabb)
    GeometryDecoder(n_features, isotropic)    modules/decoders.py:57-110   (net.0, net.2, xyz_offsets, scales.0, scales.2, rotations.0)
    AppearanceDecoder(n_features, ...)        modules/decoders.py:16-54    (net.0, net.2, opacity, shs)
    decode_attributes(...)                    SinGS.get_gs_attrs, sings_hybrid.py:249-313 (single level)

Tri-plane sampling (forward, plane / coordinate gradients) and every bias + activation (forward, backward, bias
gradient) are HIP kernels behind the C ABI (sg_triplane_*, sg_bias_act_*); the GEMMs are library GEMMs (torch.mm ->
rocBLAS / hipBLASLt).  No CPU fallback.
"""
import ctypes as C
import itertools

import torch
import torch.nn as nn

from . import _lib
from .rasterizer import _ptr

ACT_NONE, ACT_GELU, ACT_SIGMOID, ACT_SOFTPLUS_REF = 0, 1, 2, 3


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _is_feature_minor(p):
    """[1, F, H, W] in channels_last memory format = [H][W][F] in memory (and not ALSO plain-contiguous: H W > 1, F > 1)."""
    return (p.dim() == 4 and p.dtype == torch.float32 and p.is_contiguous(memory_format=torch.channels_last)
            and not p.is_contiguous())


def _tp_struct(grids, aabb, keep):
    tp = _lib.SgTriplane()
    tp.n_scales, tp.feat = len(grids), int(grids[0][0].shape[1])
    # feature-minor planes (HexPlaneField(feature_minor=True)) are used in place by the kernels: no texel-major copy per call
    tp.feature_minor = int(all(_is_feature_minor(p) for planes in grids for p in planes))
    for s, planes in enumerate(grids):
        # plane (0,1) is [1, F, Ry, Rx], (0,2) is [1, F, Rz, Rx], (1,2) is [1, F, Rz, Ry]
        rx, ry, rz = int(planes[0].shape[3]), int(planes[0].shape[2]), int(planes[1].shape[2])
        tp.res[s][0], tp.res[s][1], tp.res[s][2] = rx, ry, rz
        for c in range(3):
            t = planes[c].detach() if tp.feature_minor else planes[c].detach().contiguous().float()
            keep.append(t)
            tp.planes[s][c] = t.data_ptr()
    a = _aabb_host(aabb)
    for r in range(2):
        for k in range(3):
            tp.aabb[r][k] = a[r][k]
    return tp


import weakref

_aabb_cache = {}


def _aabb_host(aabb):
    """Host copy of the (non-trainable) bounding box, read back once per VALUE: a D2H copy in every forward and backward
    was a host synchronisation twice per training step.  An entry belongs to ONE tensor object (weak reference + version
    counter): a different tensor that later lands at the same address (a second HexPlaneField after the first was freed;
    a fresh tensor's version is 0 too) never sees the old bounds."""
    slot, key = (aabb.data_ptr(), str(aabb.device)), aabb._version
    hit = _aabb_cache.get(slot)
    if hit is None or hit[0] != key or hit[2]() is not aabb:
        a = aabb.detach().float().cpu()
        hit = (key, [[float(a[r, k]) for k in range(3)] for r in range(2)], weakref.ref(aabb))
        if len(_aabb_cache) > 64:
            _aabb_cache.clear()
        _aabb_cache[slot] = hit
    return hit[1]


# ---- the point-only half of the tri-plane backward, early -------------------------------------------------------------------
# A quarter of the tri-plane backward (texel-major planes, zeroed gradient planes, three counting sorts of the points: ~120 us at
# 150 k points, all latency) needs neither dL/dfeats nor anything else the backward pass produces: it is launched from the FORWARD on
# a side stream, where the decoders' forward hides it, and the backward only waits for its event.  On by default outside HIP-graph
# capture; inside a capture (where an unjoined side stream is an error if backward() never runs) only after
# `prepare_triplane_backward_early(True)`; `False` turns it off everywhere.
_TP = {"mode": None, "streams": {}, "late": False}
_TP_PENDING = []


def flush_triplane_prepare(stream=None):
    """Issue a tri-plane backward preparation that `prepare_triplane_backward_late(True)` held back (``stream``: on that stream
    instead of the module's own side stream)."""
    while _TP_PENDING:
        fn = _TP_PENDING.pop(0)
        fn() if stream is None else fn(stream)


def prepare_triplane_backward_late(flag=True):
    """Hold the early preparation back until `flush_triplane_prepare()` (or the backward pass): beside the first decoder layers its
    counting sorts cost the forward chain more than they save (sings_amd.train_step issues it at the end of the forward)."""
    _TP["late"] = bool(flag)


def prepare_triplane_backward_early(flag=True):
    _TP["mode"] = bool(flag)


def _tp_early(dev):
    if _TP["mode"] is False:
        return False
    return True if _TP["mode"] else not torch.cuda.is_current_stream_capturing()


# ---- capture order: the critical chain first ---------------------------------------------------------------------------------
# ROCm 7.2's graph executor gives every kernel node a queue while it walks the captured graph (DEBUG_HIP_GRAPH_DOT_PRINT=1 prints
# the assignment): the FIRST-captured dependent of a node stays on its queue, the k-th one goes to queue (q + k) mod 4 behind a
# signal; lists that share a queue run in list order, not in time order.  A side stream forked BEFORE the next kernel of the main
# chain is captured therefore KEEPS the main chain's queue and sends the main chain to another one: in the round-4 step each of the
# ten input-gradient kernels of the decoders' backward sat on a different queue than its predecessor (a signal hop each) and two of
# them behind a weight-gradient kernel of the side stream that happened to share their queue.  The weight-gradient launches are
# therefore DEFERRED by one kernel of the main chain: kept as a closure whose wait is an event recorded at the fork point (nothing
# starts later on the GPU) and issued right after the next input-gradient kernel (2.18 -> 2.06 ms per step, profiles/r05_graph_queues.log).
_DEFER = []


def _defer_side(fn):
    _DEFER.append(fn)


def _flush_deferred():
    try:
        while _DEFER:
            _DEFER.pop(0)()
    except BaseException:
        del _DEFER[:]
        raise


# ---- gradient arena ----------------------------------------------------------------------------------------------------------
# Frame-parallel training reduces the parameter gradients with ONE collective over ONE flat buffer.  Instead of
# gathering the .grad tensors into that buffer and scattering them back every step (two extra passes over 35 MB), the kernels that
# PRODUCE the large gradients -- the tri-plane scatter (33 of the 35 MB) and the decoders' weight gradients -- write them straight
# into the caller's buffer: `set_gradient_arena(params, flat)` registers a slot per parameter, the backward functions below hand
# autograd a fresh view of the slot (AccumulateGrad adopts it without a copy when .grad is None, the precondition the training
# step already guarantees), so `p.grad` IS a piece of `flat`.  Gradients produced elsewhere (biases, anchors) are small; the
# caller copies those into their slots (`arena_sync`).
_ARENA = {}                            # parameter storage address -> [flat, offset, shape, weakref(parameter), weakref(last view handed out)]


_ARENA_REG = {"params": None, "stale": False}       # the registered parameter objects (weak) / dropped by a topology change


def invalidate_gradient_arena():
    """The set of parameters changed (AvatarStep.set_topology): the registration describes parameters that no longer exist.  Nothing is
    written into the old buffer any more, and ``arena_sync`` raises until ``set_gradient_arena`` has registered the new set."""
    if _ARENA or _ARENA_REG["params"] is not None:
        _ARENA.clear()
        _ARENA_REG["stale"] = True


def set_gradient_arena(params, flat):
    """Register (or, with ``params=None``, drop) the flat gradient buffer.  Returns the per-parameter views of ``flat``."""
    _ARENA.clear()
    _ARENA_REG["stale"] = False
    _ARENA_REG["params"] = None if params is None else [weakref.ref(p) for p in params]
    if params is None:
        return []
    views, o = [], 0
    for p in params:
        n = p.numel()
        if not (p.is_contiguous() or _is_feature_minor(p)):
            raise ValueError("gradient arena: parameters must be contiguous (or feature-minor tri-plane planes)")
        _ARENA[p.data_ptr()] = [flat, o, tuple(p.shape), weakref.ref(p), None]
        views.append(_like_layout(flat[o:o + n], p))             # (a feature-minor plane's slot holds its gradient feature-minor too)
        o += n
    if o != flat.numel() or flat.dtype != torch.float32:
        raise ValueError("gradient arena: `flat` must be fp32 with exactly sum(p.numel()) elements")
    return views


def _like_layout(flat_slice, param_like):
    """A view of ``flat_slice`` (param_like.numel() elements) with the SHAPE and the memory layout of ``param_like``: plain for a
    contiguous parameter, [1,H,W,F].permute(0,3,1,2) for a feature-minor one (channels_last)."""
    if _is_feature_minor(param_like):
        n, f, h, w = param_like.shape
        return flat_slice.view(n, h, w, f).permute(0, 3, 1, 2)
    return flat_slice.view(param_like.shape)


def _arena_out(param_like):
    """A FRESH view of the arena slot of the parameter `param_like` aliases (autograd adopts a gradient tensor without copying
    only if nobody else holds that tensor object), or a new tensor when there is no arena / no slot."""
    hit = _ARENA.get(param_like.data_ptr())
    if hit is None or hit[2] != tuple(param_like.shape) or hit[0].device != param_like.device:
        return torch.empty_like(param_like, dtype=torch.float32)
    flat, o, shape, pref, last = hit
    # The slot is the memory `p.grad` lives in once autograd has adopted the view.  Writing the NEXT gradient there is only
    # right when nothing accumulates: p.grad must be None (zero_grad(set_to_none=False), micro-batch accumulation: the kernel
    # would overwrite p.grad's own memory and AccumulateGrad would then add the slot to itself -- 2 x the last gradient instead of
    # old + new), and the view of an earlier hand-out must be gone (a weight or plane used twice in ONE graph: the first use's
    # gradient still sits in the slot, waiting to be accumulated).  Otherwise: a fresh tensor, autograd adds, arena_sync copies.
    p = pref()
    if p is None or p.grad is not None or (last is not None and last() is not None):
        return torch.empty_like(param_like, dtype=torch.float32)
    v = _like_layout(flat[o:o + param_like.numel()], param_like)
    hit[4] = weakref.ref(v)
    return v


def arena_sync(params, views, to_arena=True):
    """Gradients that did not land in their slot (produced by torch's own backward functions): copy them in (before the
    collective) / back out (after it) with one multi-tensor launch."""
    reg = _ARENA_REG["params"]
    if _ARENA_REG["stale"] or reg is None or len(reg) != len(params) or any(r() is not p for r, p in zip(reg, params)):
        raise RuntimeError("gradient arena: the registration is stale (a densify / prune replaced parameters, or another parameter list "
                           "is passed than the one registered): call set_gradient_arena(params, flat) for the CURRENT parameters first")
    if any(tuple(v.shape) != tuple(p.shape) for p, v in zip(params, views)):
        raise RuntimeError("gradient arena: the views do not match the parameters' shapes (views of an earlier registration)")
    pairs = [(p.grad, v) for p, v in zip(params, views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
    if pairs:
        if to_arena:
            torch._foreach_copy_([v for _, v in pairs], [g for g, _ in pairs])
        else:
            torch._foreach_copy_([g for g, _ in pairs], [v for _, v in pairs])
    return len(pairs)


class _Triplane(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, aabb, n_scales, *planes):
        if not pts.is_cuda:
            raise RuntimeError("sings_amd.decode: tensors must live on the GPU (no CPU fallback)")
        lib = _lib.load()
        grids = [planes[3 * s:3 * s + 3] for s in range(n_scales)]
        keep = []
        tp = _tp_struct(grids, aabb, keep)
        x = pts.detach().contiguous().float()
        dev, N = x.device, int(x.shape[0])
        ws = torch.empty(int(lib.sg_triplane_ws_bytes(C.byref(tp))), dtype=torch.uint8, device=dev)
        feats = torch.empty((N, tp.n_scales * tp.feat), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_triplane_forward(C.byref(tp), N, _ptr(x), _ptr(ws), _ptr(feats), _stream(dev)), "triplane forward")
        ctx.save_for_backward(x, aabb, *planes)
        ctx.n_scales = n_scales
        ctx.early = None
        if any(ctx.needs_input_grad) and _tp_early(dev):
            bws = torch.empty(int(lib.sg_triplane_bwd_ws_bytes(C.byref(tp), N)), dtype=torch.uint8, device=dev)
            side = _TP["streams"].get(dev.index)
            if side is None:
                side = _TP["streams"][dev.index] = torch.cuda.Stream(dev)
            cur = torch.cuda.current_stream(dev)
            # (NOT deferred like the weight gradients below: issued first, this chain keeps the forward's queue and the decoders move
            #  to the next one -- one hop; issued after the first decoder layer it shares a queue with the regularisers and the weight
            #  gradients and was executed after them, 2.14 instead of 2.06 ms per step)
            ev = torch.cuda.Event()
            fork = torch.cuda.Event()
            fork.record(cur)

            gz = []                                              # feature-minor planes: the zero-filled gradient buffer, made early too

            def launch(side=side):
                side.wait_event(fork)
                with torch.cuda.device(dev), torch.cuda.stream(side):
                    _lib.check(lib.sg_triplane_backward_prepare(C.byref(tp), N, _ptr(x), _ptr(bws), C.c_void_p(side.cuda_stream)),
                               "triplane backward (prepare)")
                    if tp.feature_minor and not any(_ARENA.get(p.data_ptr()) is not None for p in planes):
                        gz.append(torch.zeros(sum(p.numel() for p in planes), dtype=torch.float32, device=dev))
                    ev.record(side)
                bws.record_stream(side); x.record_stream(side)
                for p in planes:
                    p.record_stream(side)
            if _TP.get("late"):
                _TP_PENDING.append(launch)                       # issued by flush_triplane_prepare() / the backward
            else:
                launch()
            ctx.early = (bws, ev, gz)
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        lib = _lib.load()
        x, aabb, *planes = ctx.saved_tensors
        n_scales = ctx.n_scales
        grids = [planes[3 * s:3 * s + 3] for s in range(n_scales)]
        keep = []
        tp = _tp_struct(grids, aabb, keep)
        dev, N = x.device, int(x.shape[0])
        flush_triplane_prepare()
        early = ctx.early
        ws = early[0] if early else torch.empty(int(lib.sg_triplane_bwd_ws_bytes(C.byref(tp), N)), dtype=torch.uint8, device=dev)
        cur = torch.cuda.current_stream(dev)
        if tp.feature_minor:
            # gradient planes feature-minor like the parameters, written in place by the scatter; they must start from zero
            if early and early[2]:
                gflat = early[2][0]                              # zero-filled on the side stream with the preparation
                gflat.record_stream(cur)
                dplanes, o = [], 0
                for p in planes:
                    dplanes.append(_like_layout(gflat[o:o + p.numel()], p)); o += p.numel()
            else:
                if not any(_ARENA.get(p.data_ptr()) is not None for p in planes):
                    gflat = torch.zeros(sum(p.numel() for p in planes), dtype=torch.float32, device=dev)      # ONE fill
                    dplanes, o = [], 0
                    for p in planes:
                        dplanes.append(_like_layout(gflat[o:o + p.numel()], p)); o += p.numel()
                else:                                            # (a registered arena: its slots, or fresh tensors where a slot is busy)
                    dplanes = [_arena_out(p) for p in planes]
                    for d in dplanes:
                        d.zero_()
        else:
            dplanes = [_arena_out(p) for p in planes]                # (straight into the caller's flat buffer, if registered)
        arr = ((C.c_void_p * 3) * 4)()
        for s in range(n_scales):
            for c in range(3):
                arr[s][c] = dplanes[3 * s + c].data_ptr()
        dxyz = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        df = dfeats.contiguous().float()
        with torch.cuda.device(dev):
            if early:
                cur.wait_event(early[1])                         # the prepared half (launched by the forward)
                ctx.early = None
                fn = lib.sg_triplane_backward_prepared
            else:
                fn = lib.sg_triplane_backward
            _lib.check(fn(C.byref(tp), N, _ptr(x), _ptr(ws), _ptr(df), C.byref(arr), _ptr(dxyz), _stream(dev)), "triplane backward")
        _flush_deferred()                                        # (the first layers' weight gradients: behind this kernel)
        return (dxyz, None, None) + tuple(dplanes)


def init_grid_param(grid_nd, in_dim, out_dim, reso, a=0.1, b=0.5, device='cuda'):
    """hexplane.py:18-43 (3-D inputs only)."""
    assert in_dim == len(reso) == 3 and grid_nd == 2
    grid_coefs = nn.ParameterList()
    for coo_comb in itertools.combinations(range(in_dim), grid_nd):
        p = nn.Parameter(torch.empty([1, out_dim] + [reso[cc] for cc in coo_comb[::-1]], device=device))
        nn.init.uniform_(p, a=a, b=b)
        grid_coefs.append(p)
    return grid_coefs


class HexPlaneField(nn.Module):
    def __init__(self, planeconfig, bounds=1., device='cuda', feature_minor=False):
        """``feature_minor=True``: the planes keep the reference's SHAPE [1, F, H, W] (checkpoints load and save unchanged) but live
        in torch's channels_last memory format, i.e. [H][W][F] -- the layout the sampling kernels want -- so that neither the
        parameters nor their gradients are re-laid-out per call (one 18-us and one 28-us copy of the 33 MB of planes forward /
        in the backward's preparation, one 23-us copy of the gradients back, per training step at 150 k points)."""
        super().__init__()
        aabb = torch.tensor([[bounds, bounds, bounds], [-bounds, -bounds, -bounds]], dtype=torch.float32)
        self.aabb = nn.Parameter(aabb, requires_grad=False).to(device)
        self.grid_config = [planeconfig]
        self.multiscale_res_multipliers = planeconfig["multires"]
        self.concat_features = True
        self.grids = nn.ModuleList()
        self.feat_dim = 0
        for res in self.multiscale_res_multipliers:
            config = dict(self.grid_config[0])
            config["resolution"] = [r * res for r in config["resolution"][:3]]
            gp = init_grid_param(config["grid_dimensions"], config["input_coordinate_dim"], config["output_coordinate_dim"],
                                 config["resolution"], device=device)
            self.feat_dim += gp[-1].shape[1]
            if feature_minor:
                for p in gp:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
            self.grids.append(gp)

    @property
    def get_aabb(self):
        return self.aabb[0], self.aabb[1]

    def set_aabb(self, xyz_max, xyz_min):
        self.aabb = nn.Parameter(torch.tensor([xyz_max, xyz_min], dtype=torch.float32, device=self.aabb.device), requires_grad=False)

    def forward(self, pts, timestamps=None):
        assert timestamps is None, "spatial planes only (the reference's human model never passes timestamps)"
        planes = [p for gp in self.grids for p in gp]
        return _Triplane.apply(pts.reshape(-1, 3), self.aabb, len(self.grids), *planes)

    get_density = forward


def _lin_fwd(x, W, b, act, row_offset):
    """h = act(x W^T + b) in one kernel (sg_linear_forward); returns (h, aux, Wc): aux is what the backward needs besides x --
    gelu'(z) for GELU (so the backward's act' is one multiplication), the output h for the sigmoid, z for the softplus."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("sings_amd.decode: tensors must live on the GPU (no CPU fallback)")
    Wc = W.contiguous().float()
    bc = None if b is None else b.contiguous().float()
    dev, N, Cout, Cin = x.device, int(x.shape[0]), int(W.shape[0]), int(W.shape[1])
    ro = row_offset.contiguous().float().reshape(-1) if row_offset is not None else None
    h = torch.empty((N, Cout), dtype=torch.float32, device=dev)
    aux = torch.empty_like(h) if act in (ACT_GELU, ACT_SOFTPLUS_REF) else None
    with torch.cuda.device(dev):
        _lib.check(lib.sg_linear_forward(N, Cin, Cout, act, _ptr(x), _ptr(Wc), _ptr(bc), _ptr(ro), _ptr(aux), _ptr(h),
                                         _stream(dev)), "linear forward")
    if act == ACT_SIGMOID:
        aux = h
    return h, aux, Wc


# ---- weight gradients beside the backward chain -----------------------------------------------------------------------------
# dW = dz^T x is needed by nobody downstream in the backward pass, and sg_weight_grad is bound by the matrix cores (one workgroup
# per CU) while the input-gradient kernels of the layers in front of it are bound by HBM: run on a second stream it fills the idle
# half of either.  Opt-in (`overlap_weight_grads(True)`): the gradients are complete when backward() RETURNS (the streams join in
# an end-of-backward callback), so parameters must not carry a .grad that autograd would add to DURING the pass -- true for the
# usual `optimizer.zero_grad()` (set_to_none) loop, not for gradient accumulation over several backward() calls.
_WG = {"on": False, "streams": {}, "armed": False}


def overlap_weight_grads(flag=True):
    """Run the decoders' weight-gradient kernels on a side stream, concurrently with the rest of the backward pass."""
    _WG["on"] = bool(flag)


def _wg_join():
    _WG["armed"] = False
    _flush_deferred()
    for dev_index, side in _WG["streams"].items():
        torch.cuda.current_stream(torch.device("cuda", dev_index)).wait_stream(side)


def reset_deferred():
    """Drop what a backward pass that RAISED left behind: deferred side-stream launches (their closures hold raw pointers to gradient
    tensors that no parameter adopted), a held-back tri-plane preparation, the armed end-of-backward join.  The step that raised is
    lost; the next one starts clean (sings_amd.train_step.AvatarStep.backward calls this on any exception)."""
    del _DEFER[:]
    del _TP_PENDING[:]
    _WG["armed"] = False


def _wg_stream(dev):
    side = _WG["streams"].get(dev.index)
    if side is None:
        side = _WG["streams"][dev.index] = torch.cuda.Stream(dev)
    if not _WG["armed"]:
        del _DEFER[:]                                            # (nothing of an earlier pass may be pending when a new one arms the join)
        torch.autograd.Variable._execution_engine.queue_callback(_wg_join)     # joins when this backward pass ends
        _WG["armed"] = True
    return side


def _lin_bwd(x, W, aux, act, has_b, dh, need_dx, need_dw, dx_into=None):
    """dz = dh * act'(z) in the prologue of dx = dz W (sg_linear_backward), then dW = dz^T x, db = column sums (sg_weight_grad).
    ``dx_into``: an existing [N,Cin] gradient this layer ADDS its dx to (sg_linear_backward_accumulate)."""
    lib = _lib.load()
    dev, N, Cout, Cin = x.device, int(x.shape[0]), int(W.shape[0]), int(W.shape[1])
    dh = dh.contiguous().float()
    dz = torch.empty_like(dh) if act != ACT_NONE else dh
    dx = None
    with torch.cuda.device(dev):
        if need_dx:
            acc = dx_into is not None
            dx = dx_into if acc else torch.empty((N, Cin), dtype=torch.float32, device=dev)
            fn = lib.sg_linear_backward_accumulate if acc else lib.sg_linear_backward
            _lib.check(fn(N, Cin, Cout, act, _ptr(aux) if act != ACT_NONE else None, None, _ptr(dh), _ptr(W),
                          _ptr(dz) if act != ACT_NONE else None, _ptr(dx), _stream(dev)), "linear backward")
        elif act == ACT_GELU:                                   # a layer whose input needs no gradient: only dz = dh * gelu'(z)
            torch.mul(dh, aux, out=dz)
        elif act == ACT_SIGMOID:
            torch.mul(dh, aux * (1.0 - aux), out=dz)
        elif act == ACT_SOFTPLUS_REF:
            torch.mul(dh, torch.sigmoid(aux), out=dz)
        dW = db = None
        if need_dw:
            dW, db = _weight_grad(x, W, dz, has_b)
    return dx, dW, db


def _weight_grad(x, W, dz, has_b, flush=True):
    """dW = dz^T x and db = column sums of dz in one pass on the matrix cores (sg_weight_grad); on the side stream, one kernel of the
    main chain late, when ``overlap_weight_grads`` is on.  Call it right AFTER the kernel that produced dz has been issued
    (``flush=False``: a further layer of the same kernel -- the launches deferred so far stay deferred)."""
    lib = _lib.load()
    dev, N, Cout, Cin = x.device, int(x.shape[0]), int(W.shape[0]), int(W.shape[1])
    with torch.cuda.device(dev):
        dW = _arena_out(W)
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_b else None
        ws2 = torch.empty(int(lib.sg_weight_grad_ws_bytes(N, Cout, Cin)), dtype=torch.uint8, device=dev)
        # (a parameter that already carries a .grad -- gradient accumulation, zero_grad(set_to_none=False) -- makes AccumulateGrad READ dW
        #  on the main stream at once: the deferred launch would write memory autograd has read and may have freed.  Inline there.)
        if _WG["on"] and getattr(W, "grad", None) is None:
            side, cur = _wg_stream(dev), torch.cuda.current_stream(dev)
            fork = torch.cuda.Event()
            fork.record(cur)                                     # dz (and x) are complete on the backward stream
            if flush:
                _flush_deferred()                                # the previous layer's: this layer's dx kernel has been issued

            # (the closure must not hold dW / db: AccumulateGrad adopts a returned gradient without a copy only while nobody else
            #  references it -- and a copy made BEFORE the deferred kernel has written it would be a copy of nothing.  p.grad
            #  keeps them alive until the streams have joined: _wg_join)
            p_dW, p_db = _ptr(dW), _ptr(db)

            def launch(dz=dz, x=x, ws2=ws2):
                side.wait_event(fork)
                with torch.cuda.device(dev), torch.cuda.stream(side):
                    _lib.check(lib.sg_weight_grad(N, Cout, Cin, _ptr(dz), _ptr(x), _ptr(ws2), p_dW, p_db,
                                                  C.c_void_p(side.cuda_stream)), "weight gradient")
                for t in (dz, x, ws2):
                    t.record_stream(side)
            _defer_side(launch)
        else:
            _lib.check(lib.sg_weight_grad(N, Cout, Cin, _ptr(dz), _ptr(x), _ptr(ws2), _ptr(dW), _ptr(db), _stream(dev)),
                       "weight gradient")
    return dW, db


class _LinearAct(torch.autograd.Function):
    """h = act(x @ W^T + b) as ONE kernel on the matrix cores (sg_linear_forward: bias + activation in the GEMM's epilogue);
    backward: sg_linear_backward (dz = dh * act' in the prologue of dx = dz W, dz stored once) + sg_weight_grad.
    No library GEMM, no separate element-wise pass (round 1: torch.addmm / torch.mm + sg_bias_act_*)."""

    @staticmethod
    def forward(ctx, x, W, b, act, row_offset):
        x = x.contiguous().float()
        h, aux, Wc = _lin_fwd(x, W, b, act, row_offset)
        ctx.save_for_backward(x, Wc, aux if aux is not None else torch.empty(0, device=x.device))
        ctx.act, ctx.has_b = act, b is not None
        return h

    @staticmethod
    def backward(ctx, dh):
        x, W, aux = ctx.saved_tensors
        need_dw = ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])
        dx, dW, db = _lin_bwd(x, W, aux, ctx.act, ctx.has_b, dh, ctx.needs_input_grad[0], need_dw)
        return dx, dW, db, None, None


def linear_act(x, lin, act=ACT_NONE, row_offset=None):
    return _LinearAct.apply(x, lin.weight, lin.bias, act, row_offset)


class _LinearActShared(torch.autograd.Function):
    """A layer that shares its INPUT with other layers of the same `slot` (the tri-plane features read by both decoders' first
    layers, sings_hybrid.py:259-262) without tying their backward passes together: whichever layer's backward runs first writes the
    input gradient (and hands it to autograd), the others ADD theirs into that very tensor (sg_linear_backward_accumulate) and hand
    autograd nothing -- one [N,Cin] gradient, no addition kernel, and each layer's backward runs as soon as ITS gradient is there.
    (`linear_fan` needs all of them at once: in the training step the appearance decoder's first layer then waited for the geometry
    decoder's, i.e. for the k-NN regulariser; here its 80 us run in that wait.)"""

    @staticmethod
    def forward(ctx, x, W, b, act, slot):
        x = x.contiguous().float()
        h, aux, Wc = _lin_fwd(x, W, b, act, None)
        ctx.save_for_backward(x, Wc, aux if aux is not None else torch.empty(0, device=x.device))
        ctx.act, ctx.has_b, ctx.slot = act, b is not None, slot
        return h

    @staticmethod
    def backward(ctx, dh):
        x, W, aux = ctx.saved_tensors
        need_dw = ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])
        need_dx = ctx.needs_input_grad[0]
        first = ctx.slot.get("dx") is None
        can_acc = (int(x.shape[1]) & 31) == 0
        if need_dx and not first and can_acc:
            _, dW, db = _lin_bwd(x, W, aux, ctx.act, ctx.has_b, dh, True, need_dw, dx_into=ctx.slot["dx"])
            dx = None                                            # (already inside the tensor the first layer handed over)
        else:
            dx, dW, db = _lin_bwd(x, W, aux, ctx.act, ctx.has_b, dh, need_dx, need_dw)
            if need_dx and first and can_acc:
                slot = ctx.slot
                slot["dx"] = dx
                # The shared gradient belongs to THIS backward pass only.  A later pass over the same graph (retain_graph=True, one
                # autograd.grad call per loss term, a pass that reaches only one of the decoders) must start with an empty slot: it
                # would otherwise add into the earlier pass's tensor, hand autograd nothing, and lose its input gradient.
                torch.autograd.Variable._execution_engine.queue_callback(lambda: slot.pop("dx", None))
        return dx, dW, db, None, None


def linear_act_shared(x, lin, act, slot):
    return _LinearActShared.apply(x, lin.weight, lin.bias, act, slot)


class _LinearFan(torch.autograd.Function):
    """Several layers reading the SAME activation (the decoder trunk and its heads, decoders.py:41-49, 75-94; the tri-plane
    features and the two decoders, sings_hybrid.py:259-262): forward = one sg_linear_forward per layer; backward: the first
    consumer writes dx, the others ADD theirs into the same array (sg_linear_backward_accumulate) -- one [N,Cin] gradient
    instead of one per consumer plus autograd's element-wise additions (two 77-MB passes per trunk at 150 k points)."""

    @staticmethod
    def forward(ctx, x, acts, row_offsets, *wb):
        x = x.contiguous().float()
        outs, saved = [], [x]
        for i, act in enumerate(acts):
            h, aux, Wc = _lin_fwd(x, wb[2 * i], wb[2 * i + 1], act, row_offsets[i])
            outs.append(h)
            saved += [Wc, aux if aux is not None else torch.empty(0, device=x.device)]
        ctx.save_for_backward(*saved)
        ctx.acts, ctx.has_b = tuple(acts), tuple(wb[2 * i + 1] is not None for i in range(len(acts)))
        return tuple(outs)

    @staticmethod
    def _fan_plan(ctx, dhs, x):
        """Layer 0 wide and every other layer a narrow head (<= 16 columns together, at most two, act none | sigmoid): their input
        gradients in ONE kernel (sg_linear_backward_fan).  None: the per-layer path."""
        n = len(ctx.acts)
        if not (2 <= n <= 3) or any(d is None for d in dhs) or (int(x.shape[1]) & 31) or int(x.shape[0]) > 3_000_000:
            return None                                          # (the kernel's 32-bit byte offsets: N + look-ahead rows < 2^31 / 576)
        couts = [int(ctx.saved_tensors[1 + 2 * i].shape[0]) for i in range(n)]
        if couts[0] & 3 or sum(couts[1:]) > 16 or any(a not in (ACT_NONE, ACT_SIGMOID) for a in ctx.acts[1:]):
            return None
        return couts

    @staticmethod
    def backward(ctx, *dhs):
        x = ctx.saved_tensors[0]
        need_dx = ctx.needs_input_grad[0]
        can_acc = (int(x.shape[1]) & 31) == 0
        couts = _LinearFan._fan_plan(ctx, dhs, x) if (need_dx and _FAN["on"]) else None
        if couts is not None:
            return _LinearFan._backward_fused(ctx, dhs, x, couts)
        dx, grads = None, []
        for i, act in enumerate(ctx.acts):
            W, aux = ctx.saved_tensors[1 + 2 * i], ctx.saved_tensors[2 + 2 * i]
            need_dw = ctx.needs_input_grad[3 + 2 * i] or (ctx.has_b[i] and ctx.needs_input_grad[4 + 2 * i])
            if dhs[i] is None:
                grads += [None, None]
                continue
            dxi, dW, db = _lin_bwd(x, W, aux, act, ctx.has_b[i], dhs[i], need_dx, need_dw,
                                   dx_into=dx if (can_acc and dx is not None) else None)
            if need_dx:
                dx = dxi if (dx is None or can_acc) else dx + dxi
            grads += [dW, db]
        return (dx, None, None, *grads)


    @staticmethod
    def _backward_fused(ctx, dhs, x, couts):
        lib = _lib.load()
        dev, N, Cin = x.device, int(x.shape[0]), int(x.shape[1])
        n = len(couts)
        W = [ctx.saved_tensors[1 + 2 * i] for i in range(n)]
        aux = [ctx.saved_tensors[2 + 2 * i] for i in range(n)]
        dh = [d.contiguous().float() for d in dhs]
        dz = [torch.empty_like(dh[i]) if ctx.acts[i] != ACT_NONE else dh[i] for i in range(n)]
        dx = torch.empty((N, Cin), dtype=torch.float32, device=dev)
        side = (_lib.SgLinearSide * 2)()
        for j in range(1, n):
            sd = side[j - 1]
            sd.cout, sd.act = couts[j], ctx.acts[j]
            sd.aux = aux[j].data_ptr() if ctx.acts[j] != ACT_NONE else None
            sd.dh, sd.W = dh[j].data_ptr(), W[j].data_ptr()
            sd.dz_out = dz[j].data_ptr() if ctx.acts[j] != ACT_NONE else None
        with torch.cuda.device(dev):
            _lib.check(lib.sg_linear_backward_fan(N, Cin, couts[0], ctx.acts[0], _ptr(aux[0]) if ctx.acts[0] != ACT_NONE else None,
                                                  _ptr(dh[0]), _ptr(W[0]), _ptr(dz[0]) if ctx.acts[0] != ACT_NONE else None, _ptr(dx),
                                                  side, _stream(dev)), "linear backward (fan)")
        grads = []
        for i in range(n):
            need_dw = ctx.needs_input_grad[3 + 2 * i] or (ctx.has_b[i] and ctx.needs_input_grad[4 + 2 * i])
            dW, db = _weight_grad(x, W[i], dz[i], ctx.has_b[i], flush=not grads) if need_dw else (None, None)
            grads += [dW, db]
        return (dx, None, None, *grads)


_FAN = {"on": True}


def fuse_fan_backward(flag=True):
    """The narrow heads' input gradients inside the wide layer's kernel (default on; off: one accumulate pass per head)."""
    _FAN["on"] = bool(flag)


def linear_fan(x, layers):
    """layers: [(nn.Linear, act, row_offset | None), ...] -> tuple of outputs, one per layer, all reading x."""
    wb = []
    for lin, _, _ in layers:
        wb += [lin.weight, lin.bias]
    return _LinearFan.apply(x, tuple(a for _, a, _ in layers), tuple(r for _, _, r in layers), *wb)


class AppearanceDecoder(nn.Module):
    def __init__(self, n_features, hidden_dim=64, act='gelu', fixed_opacity=False):
        super().__init__()
        assert act == 'gelu', "the reference's default (and only configured) activation"
        self.hidden_dim = hidden_dim
        self.net = nn.Sequential(nn.Linear(n_features, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU())
        if not fixed_opacity:
            self.opacity = nn.Linear(hidden_dim, 1)
            self.opacity_act = nn.Sigmoid()
            self.opacity_offset = 0
        self.fixed_opacity = fixed_opacity
        self.shs = nn.Linear(hidden_dim, 16 * 3)

    def _trunk(self, x):
        return linear_act(linear_act(x, self.net[0], ACT_GELU), self.net[2], ACT_GELU)

    def reset_opacity(self, x):
        with torch.no_grad():
            o = linear_act(self._trunk(x), self.opacity)
            self.opacity_offset = torch.where(o > 0, torch.zeros_like(o), -o)

    def heads(self, x):
        """the heads on the trunk output x: one fan (their input gradients are accumulated in place)."""
        if not self.fixed_opacity:
            off = self.opacity_offset if torch.is_tensor(self.opacity_offset) else None
            shs, opacity = linear_fan(x, [(self.shs, ACT_NONE, None), (self.opacity, ACT_SIGMOID, off)])
        else:
            shs = linear_act(x, self.shs)
            opacity = torch.ones((x.shape[0], 1), device=x.device)
        return {'shs': shs.reshape(-1, 16, 3), 'opacity': opacity}

    def forward(self, x, first=None):
        """``first``: the output of net[0] + GELU when the caller computed it (decode_attributes fans the tri-plane features
        out to both decoders' first layers)."""
        h1 = linear_act(x, self.net[0], ACT_GELU) if first is None else first
        return self.heads(linear_act(h1, self.net[2], ACT_GELU))


class GeometryDecoder(nn.Module):
    def __init__(self, n_features, isotropic=True, hidden_dim=128, act='gelu'):
        super().__init__()
        assert act == 'gelu'
        self.hidden_dim, self.isotropic = hidden_dim, isotropic
        self.net = nn.Sequential(nn.Linear(n_features, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU())
        self.xyz_offsets = nn.Linear(hidden_dim, 3)
        if not isotropic:
            self.rotations = nn.Sequential(nn.Linear(hidden_dim, 6))
        self.scales = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, 1 if isotropic else 3))

    def forward(self, x, first=None):
        h1 = linear_act(x, self.net[0], ACT_GELU) if first is None else first
        x = linear_act(h1, self.net[2], ACT_GELU)
        # the heads read the same trunk output: one fan, the wide layer first (it writes dx, the narrow heads add to it)
        layers = [(self.scales[0], ACT_GELU, None), (self.xyz_offsets, ACT_NONE, None)]
        if not self.isotropic:
            layers.append((self.rotations[0], ACT_NONE, None))
        outs = linear_fan(x, layers)
        s1, xyz_offsets = outs[0], outs[1]
        rotations = outs[2] if not self.isotropic else None
        scales_aux = linear_act(s1, self.scales[2])
        if scales_aux.shape[-1] == 1 and scales_aux.is_cuda:
            # isotropic: softplus and the two `.repeat(1, 3)` in one launch each way (sg_scales_head_*)
            scales, scales_aux = _ScalesHead.apply(scales_aux)
        else:
            # scales = log(exp(scales_aux) + 1): the activation kernel on the bias-added value (identity GEMM avoided)
            scales = _Act.apply(scales_aux, ACT_SOFTPLUS_REF)
            if scales_aux.shape[-1] == 1:
                scales_aux = scales_aux.repeat(1, 3)
                scales = scales.repeat(1, 3)
        return {'xyz_offsets': xyz_offsets, 'rotations': rotations, 'scales': scales, 'scales_aux': scales_aux}


class _ScalesHead(torch.autograd.Function):
    """z [N,1] -> (scales [N,3] = softplus(z) three times, scales_aux [N,3] = z three times), decoders.py:88-98."""

    @staticmethod
    def forward(ctx, z):
        lib = _lib.load()
        z = z.contiguous().float()
        dev, N = z.device, int(z.shape[0])
        scales = torch.empty((N, 3), dtype=torch.float32, device=dev)
        aux = torch.empty_like(scales)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_scales_head_forward(N, _ptr(z), _ptr(scales), _ptr(aux), _stream(dev)), "scales head forward")
        ctx.save_for_backward(z)
        ctx.set_materialize_grads(False)                         # (an output nobody differentiates: None, not a zero-filled tensor)
        return scales, aux

    @staticmethod
    def backward(ctx, dscales, daux):
        lib = _lib.load()
        (z,) = ctx.saved_tensors
        dev, N = z.device, int(z.shape[0])
        if dscales is None and daux is None:
            return None
        ds = dscales.contiguous().float() if dscales is not None else None
        da = daux.contiguous().float() if daux is not None else None
        dz = torch.empty_like(z)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_scales_head_backward(N, _ptr(z), _ptr(ds), _ptr(da), _ptr(dz), _stream(dev)), "scales head backward")
        return dz


class _Act(torch.autograd.Function):
    """element-wise activation through the same kernels (no bias)."""

    @staticmethod
    def forward(ctx, z, act):
        lib = _lib.load()
        z = z.contiguous().float()
        dev, N, Cc = z.device, int(z.shape[0]), int(z.shape[1])
        h = torch.empty_like(z)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_bias_act_forward(N, Cc, act, _ptr(z), None, None, None, _ptr(h), _stream(dev)), "act forward")
        ctx.save_for_backward(z)
        ctx.act = act
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        (z,) = ctx.saved_tensors
        dev, N, Cc = z.device, int(z.shape[0]), int(z.shape[1])
        dh = dh.contiguous().float()
        dz = torch.empty_like(dh)
        ws = torch.empty(int(lib.sg_bias_act_ws_bytes(N, Cc)), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_bias_act_backward(N, Cc, ctx.act, _ptr(z), None, _ptr(dh), _ptr(ws), _ptr(dz), None, _stream(dev)),
                       "act backward")
        return dz, None


def decode_attributes(xyz, triplane, geometry_dec, appearance_dec, thickness_factor=1.0, scaling_multiplier=None,
                      geometry_hook=None):
    """SinGS.get_gs_attrs for one level (sings_hybrid.py:249-313): the dict the LBS/render path consumes.
    ``geometry_hook(xyz_canon, scales) -> (xyz_canon, scales)``: applied to the geometry decoder's outputs BEFORE the appearance
    decoder is issued -- autograd nodes the hook creates are visited after the appearance decoder's in the backward pass
    (sings_amd.train_step uses it to add the k-NN regulariser's gradient as late as possible)."""
    tri_feats = triplane(xyz)
    # both decoders' first layers read the tri-plane features: ONE feature gradient (the second layer to run its backward adds into
    # the first one's), but two autograd nodes -- the appearance decoder's backward does not wait for the geometry decoder's
    if tri_feats.is_cuda and torch.is_grad_enabled() and (int(tri_feats.shape[1]) & 31) == 0:
        slot = {}
        g1 = linear_act_shared(tri_feats, geometry_dec.net[0], ACT_GELU, slot)
        a1 = linear_act_shared(tri_feats, appearance_dec.net[0], ACT_GELU, slot)
    else:
        g1, a1 = linear_fan(tri_feats, [(geometry_dec.net[0], ACT_GELU, None), (appearance_dec.net[0], ACT_GELU, None)])
    g = geometry_dec(tri_feats, first=g1)
    scales = g['scales']
    if thickness_factor != 1.0:
        scales = torch.cat([scales[:, :-1], scales[:, -1:] * thickness_factor], dim=1)
    if scaling_multiplier is not None:
        scales = scales * scaling_multiplier
    xyz_canon = xyz + g['xyz_offsets']
    if geometry_hook is not None:
        xyz_canon, scales = geometry_hook(xyz_canon, scales)
    a = appearance_dec(tri_feats, first=a1)
    return {"xyz_canon": xyz_canon, "xyz_offsets": g['xyz_offsets'], "rot6d_canon": g['rotations'],
            "scales_aux": g['scales_aux'], "scales": scales, "opacity": a['opacity'], "shs": a['shs']}
