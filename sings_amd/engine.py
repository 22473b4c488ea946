"""Pre-allocated forward/backward driver over the C ABI (no autograd, no per-call allocation).

This is what a training loop that owns its buffers (and bench.py) calls: workspaces and the
flat gradient buffer are sized once for 288 GB-class HBM and reused every step, the capacity
for (tile, Gaussian) pairs is a fixed upper bound, and nothing synchronises inside a step.  The
gradient tensors are views into ONE flat fp32 buffer so that frame-parallel training can
all-reduce them with a single RCCL call (see sings_amd/dp.py).
"""
import ctypes as C

import torch

from . import _lib
from .rasterizer import _settings_struct, _ptr


class RasterEngine:
    # flat gradient layout (floats per Gaussian): means3D 3, scales 3, rotations 4, opacity 1, sh 3*M
    def __init__(self, P, W, H, sh_coeffs, device, capacity_pairs, grad_flat=None):
        self.lib = _lib.load()
        self.P, self.W, self.H, self.M = int(P), int(W), int(H), int(sh_coeffs)
        self.dev = torch.device(device)
        self.cap = int(capacity_pairs)
        L = _lib.layout(self.P, self.W, self.H, self.cap)
        self.L = L
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.geom = torch.empty(L.geom_bytes, **u8)
        self.binning = torch.empty(L.bin_bytes, **u8)
        self.binning[:L.bin_ranges].zero_()                  # counters zeroed ONCE; every forward leaves them zeroed
        self._clean = True                                   # (SG_FLAG_WS_CLEAN: no per-call zeroing launch)
        self.img = torch.empty(L.img_bytes, **u8)
        self.bwd_ws = torch.empty(L.bwd_bytes, **u8)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.color = torch.empty((3, self.H, self.W), **f32)
        self.radii = torch.empty((self.P,), dtype=torch.int32, device=self.dev)
        per = 3 + 3 + 4 + 1 + 3 * self.M
        # grad_flat: optional caller-owned storage (one row of a [views, P*per] buffer when several engines render the
        # views of one step concurrently and their gradients are summed in one pass)
        if grad_flat is not None and (grad_flat.numel() != self.P * per or grad_flat.dtype != torch.float32
                                      or not grad_flat.is_contiguous() or grad_flat.device != self.dev):
            raise ValueError(f"grad_flat must be a contiguous fp32 tensor of {self.P * per} elements on {self.dev}")
        self.grad_flat = torch.empty(self.P * per, **f32) if grad_flat is None else grad_flat.view(-1)
        o = 0
        def carve(n, *shape):
            nonlocal o
            v = self.grad_flat[o:o + n].view(*shape); o += n
            return v
        self.d_means3D = carve(self.P * 3, self.P, 3)
        self.d_scales = carve(self.P * 3, self.P, 3)
        self.d_rots = carve(self.P * 4, self.P, 4)
        self.d_opacity = carve(self.P, self.P, 1)
        self.d_sh = carve(self.P * 3 * self.M, self.P, self.M, 3) if self.M else None
        self.d_means2D = torch.empty((self.P, 3), **f32)      # densification statistic, not reduced as a weight grad
        self._keep = []
        self._s = None
        self._chain = None                                    # inside ViewBatch.run: callable -> (accumulate, wait-for event, done event)
        self.throughput = False                               # SG_FLAG_THROUGHPUT: set by a ViewBatch that keeps several views in flight

    def set_camera(self, raster_settings, short_lists=False, long_rows=False):
        """``short_lists``: the caller knows (from a sizing pass over this scene) that no tile list exceeds 1024 entries (what the
        compositing workgroups sort themselves); the two long-list sort launches are then skipped and, on images of many tiles, the
        frame is binned directly (SG_FLAG_SHORT_LISTS).  ``long_rows``: images of few tiles -- no list exceeds 16384 entries: direct
        binning there (SG_FLAG_LONG_ROWS).  A longer list makes ``num_rendered()`` return ``_lib.NUM_RENDERED_LONG_LIST`` (the frame
        rendered the background): call set_camera again without the hint."""
        self._keep = []
        self._s = _settings_struct(raster_settings, self.dev, self.M, self._keep)
        self._hint = (_lib.FLAG_SHORT_LISTS if short_lists else 0) | (_lib.FLAG_LONG_ROWS if long_rows else 0)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def forward(self, means3D, shs, opacities, scales, rotations, sync_num_rendered=False):
        nr = C.c_int64(-1)
        self._s.flags = self._hint | (_lib.FLAG_WS_CLEAN if self._clean else 0) | (_lib.FLAG_THROUGHPUT if self.throughput else 0)
        self._clean = False                                  # (stays False if the call below raises)
        _lib.check(self.lib.sg_rasterize_forward(
            C.byref(self._s), self.P, _ptr(means3D), _ptr(shs), None, _ptr(opacities), _ptr(scales), _ptr(rotations),
            None, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.color), _ptr(self.radii),
            0, C.byref(nr) if sync_num_rendered else None, self._stream()), "forward")
        self._clean = True
        return int(nr.value)

    def backward(self, means3D, shs, opacities, scales, rotations, dL_dcolor):
        """Composite backward (gradient records of this view), then the per-Gaussian chain rule into the gradient buffer.
        Inside a ``ViewBatch`` whose views share ONE gradient buffer (``chain``) the second half first waits for the view in
        front of this one in the step's chain and ADDS to the buffer (``sg_rasterize_backward_gaussians(accumulate=1)``)."""
        _lib.check(self.lib.sg_rasterize_backward_records(
            C.byref(self._s), self.P, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.bwd_ws),
            _ptr(dL_dcolor), self._stream()), "backward (records)")
        accumulate, after, done = self._chain() if self._chain is not None else (False, None, None)
        if after is not None:
            torch.cuda.current_stream(self.dev).wait_event(after)
        _lib.check(self.lib.sg_rasterize_backward_gaussians(
            C.byref(self._s), self.P, _ptr(means3D), _ptr(shs), None, _ptr(opacities), _ptr(scales), _ptr(rotations),
            None, _ptr(self.radii), _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.bwd_ws), int(accumulate),
            _ptr(self.d_means3D), _ptr(self.d_means2D), _ptr(self.d_sh), None, _ptr(self.d_opacity),
            _ptr(self.d_scales), _ptr(self.d_rots), None, self._stream()), "backward (gaussians)")
        if done is not None:
            done.record(torch.cuda.current_stream(self.dev))

    def num_rendered(self):
        nr = C.c_int64(0)
        _lib.check(self.lib.sg_read_num_rendered(_ptr(self.binning), C.byref(nr), self._stream()), "read R")
        return int(nr.value)

    def capture(self, means3D, shs, opacities, scales, rotations, dL_dcolor=None):
        """Record one step (forward, + backward when dL_dcolor is given) into a HIP graph and return it; ``graph.replay()``
        re-runs the whole launch sequence with one host call.  Possible because nothing in a step synchronises or
        allocates: every buffer (inputs included -- update them IN PLACE between replays) is fixed, the camera is read
        from device memory.  Pays off when the step is launch-bound (small scenes, forward-only animation)."""
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):                            # warm-up on the capture stream
            self.forward(means3D, shs, opacities, scales, rotations)
            if dL_dcolor is not None:
                self.backward(means3D, shs, opacities, scales, rotations, dL_dcolor)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.forward(means3D, shs, opacities, scales, rotations)
            if dL_dcolor is not None:
                self.backward(means3D, shs, opacities, scales, rotations, dL_dcolor)
        return g


def _carve_skinned(self, carve, with_rot, rot_width, own):
    """Gradient views of the LBS-fused engines.  Default (the reference's tensor shapes): xyz 3P, scales 3P, opacity P, sh [P,M,3],
    [rot].  ``sh_planar``: xyz, scales, opacity, [rot], then the SH gradient COEFFICIENT-major, [M,P,3] (SG_FLAG_SH_PLANAR) -- the
    kernels write only the (sh_degree + 1)^2 planes in use, so the step's gradient is the PREFIX ``grad_flat[:active_floats(deg)]``
    (10 floats per Gaussian at degree 0 instead of 55): that prefix is what is folded and all-reduced; the rest stays zero (zeroed
    here, once, when the engine owns the buffer -- a caller-owned buffer must come zeroed)."""
    P, M = self.P, self.M
    self.d_xyz = carve(P * 3, P, 3); self.d_scales = carve(P * 3, P, 3); self.d_opacity = carve(P, P, 1)
    if self.sh_planar:
        self.d_rot = carve(P * rot_width, P, rot_width) if with_rot else None
        self.d_sh = carve(P * 3 * M, M, P, 3)
        if own:
            self.d_sh.zero_()
    else:
        self.d_sh = carve(P * 3 * M, P, M, 3)
        self.d_rot = carve(P * rot_width, P, rot_width) if with_rot else None
    self._head = P * 7 + (P * rot_width if with_rot else 0)


def _active_floats(self, sh_degree):
    """Length of the prefix of ``grad_flat`` that carries the step's gradient (everything, unless ``sh_planar``)."""
    if not self.sh_planar:
        return self.grad_flat.numel()
    return self._head + (int(sh_degree) + 1) ** 2 * self.P * 3


class SkinnedEngine:
    """Same idea for the LBS-fused path (sg_skinned_forward / backward): canonical Gaussians + per-frame joint
    transforms in, image + flat canonical-Gaussian gradient buffer out.  Flat layout (floats): xyz_canon 3P,
    scales 3P, opacity P, sh 3MP, [rot_canon 9P].  dL/dA [J,16] and dL/dtransl [3] are per-frame (not all-reduced)."""

    def __init__(self, P, J, W, H, sh_coeffs, device, capacity_pairs, with_rot=False, grad_flat=None, rot_width=9, sh_planar=False):
        self.lib = _lib.load()
        self.P, self.J, self.W, self.H, self.M = int(P), int(J), int(W), int(H), int(sh_coeffs)
        self.sh_planar = bool(sh_planar)
        self.dev = torch.device(device)
        self.cap = int(capacity_pairs)
        L = _lib.layout(self.P, self.W, self.H, self.cap)
        self.L = L
        u8 = dict(dtype=torch.uint8, device=self.dev)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.geom = torch.empty(L.geom_bytes, **u8); self.binning = torch.empty(L.bin_bytes, **u8)
        self.binning[:L.bin_ranges].zero_(); self._clean = True          # SG_FLAG_WS_CLEAN, as in RasterEngine
        self.img = torch.empty(L.img_bytes, **u8); self.bwd_ws = torch.empty(L.bwd_bytes, **u8)
        self.skin_ws = torch.empty(int(self.lib.sg_skin_ws_floats(self.P)), **f32)
        self.color = torch.empty((3, self.H, self.W), **f32)
        self.radii = torch.empty((self.P,), dtype=torch.int32, device=self.dev)
        rot_width = int(rot_width)                               # 9: rotation matrices, 6: the decoder's 6-D form
        per = 3 + 3 + 1 + 3 * self.M + (rot_width if with_rot else 0)
        if grad_flat is not None and (grad_flat.numel() != self.P * per or grad_flat.dtype != torch.float32
                                      or not grad_flat.is_contiguous() or grad_flat.device != self.dev):
            raise ValueError(f"grad_flat must be a contiguous fp32 tensor of {self.P * per} elements on {self.dev}")
        self.grad_flat = torch.empty(self.P * per, **f32) if grad_flat is None else grad_flat.view(-1)
        o = 0

        def carve(n, *shape):
            nonlocal o
            v = self.grad_flat[o:o + n].view(*shape); o += n
            return v
        _carve_skinned(self, carve, with_rot, rot_width, grad_flat is None)
        self.d_means2D = torch.empty((self.P, 3), **f32)
        self.d_A = torch.empty((self.J, 16), **f32); self.d_transl = torch.empty(3, **f32)
        self._keep = []; self._s = None; self._k = None
        self._chain = None
        self.throughput = False

    def set_camera(self, raster_settings, long_rows=False):
        """``long_rows``: the caller knows (from a sizing pass) that no tile list of this avatar exceeds 16384 entries: direct binning
        (SG_FLAG_LONG_ROWS, include/sings_hip.h; a longer list: ``num_rendered()`` = NUM_RENDERED_LONG_LIST, background frame)."""
        self._keep = []
        self._s = _settings_struct(raster_settings, self.dev, self.M, self._keep)
        self._rows = _lib.FLAG_LONG_ROWS if long_rows else 0

    active_floats = _active_floats

    def set_frame(self, xyz_canon, rotmat_canon, lbs_weights, A, smpl_scale, transl, ext_tfs=None):
        from .skinned import _skin_struct
        self._kkeep = []
        self._k = _skin_struct(self.dev, xyz_canon, rotmat_canon, lbs_weights, A, smpl_scale, transl, ext_tfs, self._kkeep)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def forward(self, shs, opacities, scales, sync_num_rendered=False, posed_out=None):
        """``posed_out``: optional (xyz [P,3], quaternions [P,4], scales [P,3]) fp32 tensors that also receive the posed values
        (what SinGS.forward returns: sings_hybrid.py:400-419); the render itself never reads them back."""
        nr = C.c_int64(-1)
        self._s.flags = ((_lib.FLAG_WS_CLEAN if self._clean else 0) | (_lib.FLAG_THROUGHPUT if self.throughput else 0) |
                         (_lib.FLAG_SH_PLANAR if self.sh_planar else 0) | getattr(self, "_rows", 0))
        self._clean = False
        pxyz, pq, psc = posed_out if posed_out is not None else (None, None, None)
        _lib.check(self.lib.sg_skinned_forward(
            C.byref(self._s), self.P, C.byref(self._k), _ptr(shs), _ptr(opacities), _ptr(scales), _ptr(self.geom),
            _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.color), _ptr(self.radii), _ptr(pxyz), _ptr(pq), _ptr(psc),
            C.byref(nr) if sync_num_rendered else None, self._stream()), "skinned forward")
        self._clean = True
        return int(nr.value)

    def backward(self, shs, opacities, scales, dL_dcolor):
        """As ``RasterEngine.backward``: records of this frame, then (chained inside a ``ViewBatch``) the per-Gaussian half."""
        _lib.check(self.lib.sg_rasterize_backward_records(
            C.byref(self._s), self.P, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.bwd_ws),
            _ptr(dL_dcolor), self._stream()), "skinned backward (records)")
        accumulate, after, done = self._chain() if self._chain is not None else (False, None, None)
        if after is not None:
            torch.cuda.current_stream(self.dev).wait_event(after)
        _lib.check(self.lib.sg_skinned_backward_gaussians(
            C.byref(self._s), self.P, C.byref(self._k), _ptr(shs), _ptr(opacities), _ptr(scales), _ptr(self.radii),
            _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.bwd_ws), _ptr(self.skin_ws), int(accumulate),
            None, None, _ptr(self.d_xyz), _ptr(self.d_rot), _ptr(self.d_scales), _ptr(self.d_opacity),
            _ptr(self.d_sh), _ptr(self.d_means2D), _ptr(self.d_A), _ptr(self.d_transl), self._stream()), "skinned backward (gaussians)")
        if done is not None:
            done.record(torch.cuda.current_stream(self.dev))

    def num_rendered(self):
        nr = C.c_int64(0)
        _lib.check(self.lib.sg_read_num_rendered(_ptr(self.binning), C.byref(nr), self._stream()), "read R")
        return int(nr.value)


class ViewBatch:
    """The views (frames) of ONE optimisation step in flight on several HIP streams.

    One view is a chain of short launches: the binning kernels are latency-bound (a handful of waves) and the composite
    kernels end in tile-imbalance tails.  Views of the same step are independent until their gradients are summed, the
    library keeps no state between calls, and every engine owns its workspaces -- so the views are dealt round-robin to
    ``streams`` streams and fill each other's holes (one MI355X: +24 % views/s at cfg3, +77 % frames/s for the avatar),
    bit-identical to running them one after the other.  ``run`` forks from the caller's current stream, launches
    ``fn(v, engine)`` for every view on its stream and joins; nothing synchronises with the host.  (No further stream: the
    caller's + three rendering streams are the four hardware queues HIP schedules onto.)

    Where the sum over the views is formed (``grads`` is ``[rows, per_view]``; engine v is built on row ``v % rows``):

    * rows == streams (round 3, what bench.py uses): ONE gradient row per stream.  The first view of a stream WRITES the row,
      the later views of that stream ADD to it (``sg_*_backward_gaussians(accumulate=1)``; stream order is the only ordering
      needed), so after the join ``streams`` rows are folded instead of ``views`` rows: the serial tail of a cfg3 step of 8 views
      on 3 streams reads 3 x 47 MB instead of 8 x 47 MB (the bytes of the other five rows are read by the accumulating kernels,
      inside the overlapped part of the step).  Fixed assignment and order -> bitwise reproducible.
    * rows == 1 with several streams: every view adds to the same buffer; an event per view orders the per-Gaussian halves of
      consecutive views across the streams (no fold at all).  Measured slower than stream rows on the avatar step (the waits
      couple the streams: 3 821 vs 3 940 frames/s) and on par at cfg3; kept for memory-constrained callers.
    * rows == views: the round-2 scheme, every view writes its own row.

    Either way a ``FrameParallel`` all-reduces the summed buffer over the ranks, chunk by chunk behind the fold
    (sings_amd.dp.GradientPipeline).

        grads = ViewBatch.gradient_rows(streams, per_view_floats, device)
        engines = [RasterEngine(..., grad_flat=grads[v % streams]) for v in range(views)]     # or SkinnedEngine
        batch = ViewBatch(engines, grads, streams=streams, frame_parallel=FrameParallel())     # None on one GPU
        acc = batch.run(lambda v, e: (e.forward(...), e.backward(...)))               # summed over views and ranks -> optimiser
    """

    @staticmethod
    def gradient_rows(views, per_view, device, zero=True):
        """``zero`` (default): ``sh_planar`` engines write only the SH planes in use, and with one row the row IS the
        accumulator -- an uninitialised tail would reach the optimiser.  ``zero=False`` for engines that write every
        float of their row (the reference layout) and callers that want to skip the one-off fill."""
        alloc = torch.zeros if zero else torch.empty
        return alloc((int(views), int(per_view)), dtype=torch.float32, device=device)

    def __init__(self, engines, grads, streams=3, frame_parallel=None, chunks=4, active=None):
        from .dp import GradientPipeline
        self.engines = list(engines)
        if grads.dim() == 1:
            grads = grads.view(1, -1)
        self.rows = int(grads.shape[0])
        self.n = max(1, min(int(streams), len(self.engines)))
        if self.rows not in (1, self.n, len(self.engines)):
            raise ValueError("gradient rows must be 1, the number of streams or the number of views")
        for v, e in enumerate(self.engines):
            if e.grad_flat.data_ptr() != grads[v % self.rows].data_ptr():
                raise ValueError(f"engine {v} must be built on grad_flat = grads[{v % self.rows}]")
        self.chain = self.rows == 1 and self.n > 1 and len(self.engines) > 1      # one shared buffer across streams: events
        self.grads = grads
        self.dev = grads.device
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.n)] if self.n > 1 else []
        # the rows are folded (one row: nothing to fold) and the sum is all-reduced chunk by chunk on the caller's stream once
        # the views have joined it: sings_amd.dp.GradientPipeline
        self.pipe = GradientPipeline(grads, frame_parallel, chunks=chunks, active=active)
        self.acc = self.pipe.acc
        self._active_given = active is not None
        self._events = [torch.cuda.Event() for _ in self.engines] if self.chain else None
        for e in self.engines:                                # several views in flight: no latency-only work (SG_FLAG_THROUGHPUT)
            e.throughput = self.n > 1 or getattr(e, "K", 1) > 1

    def _link(self, v, e):
        # Resolved when the view's backward actually RUNS (engine.backward calls it once): the view adds to its row iff a
        # backward of this run has already written that row -- a forward-only view (fn skips the backward) therefore never makes
        # a later view add to an unwritten buffer (round 3 decided by the view index alone); with one row across several streams
        # the per-Gaussian half additionally waits for the last backward before it (events order only those last, short kernels).
        def resolve():
            row = v % self.rows
            acc = self._written[row]
            self._written[row] = True
            if not self.chain:
                return acc, None, None
            after, self._last_event = self._last_event, self._events[v]
            return acc, after, self._events[v]
        e._chain = resolve

    def set_active(self, sh_degree):
        """The engines' SH degree changed (``oneupSHdegree``): widen (or narrow) the prefix of the gradient rows that is folded and
        all-reduced to ``engine.active_floats(sh_degree)``.  Every rank calls it at the same step."""
        self.pipe.set_active(self.engines[0].active_floats(sh_degree))
        self._active_given = True

    def _check_active(self):
        """An ``sh_planar`` engine whose camera asks for more SH planes than the configured prefix covers would have its extra
        planes silently dropped by the fold and the collective: refuse (call ``set_active(sh_degree)``)."""
        if not self._active_given:
            return
        for e in self.engines:
            st = getattr(e, "_s", None)
            if getattr(e, "sh_planar", False) and st is not None and e.active_floats(int(st.sh_degree)) > self.pipe.n:
                raise RuntimeError(f"ViewBatch: an engine renders at SH degree {int(st.sh_degree)} but only the first {self.pipe.n} "
                                   f"floats of a gradient row are folded / all-reduced: call set_active({int(st.sh_degree)})")

    def run_unreduced(self, fn):
        """Render the views only (rows left unfolded): the caller reduces them itself, e.g. ``pipe.one_shot()``.  ``fn(v, engine)``
        runs view v's forward and AT MOST ONE backward; every row must be written by at least one backward before the rows are
        folded (checked).  Outside ``run`` the engines are plain again (a later stand-alone backward writes, never adds)."""
        self._check_active()
        self._written = [False] * self.rows
        self._last_event = None
        try:
            if not self.streams:
                for v, e in enumerate(self.engines):
                    self._link(v, e)
                    fn(v, e)
            else:
                cur = torch.cuda.current_stream(self.dev)
                for st in self.streams:
                    st.wait_stream(cur)
                for v, e in enumerate(self.engines):
                    self._link(v, e)
                    with torch.cuda.stream(self.streams[v % self.n]):
                        fn(v, e)
                for st in self.streams:
                    cur.wait_stream(st)
        finally:
            for e in self.engines:
                e._chain = None
        if any(self._written) and not all(self._written):
            raise RuntimeError("ViewBatch: a gradient row was not written by any view of this step (every stream / row needs at "
                               "least one view that runs its backward); the fold would read an unwritten buffer")

    def run(self, fn):
        self.run_unreduced(fn)
        return self.pipe.reduce()


def _frame_batch(K, camera_stride, transl_stride):
    fb = _lib.SgFrameBatch()
    fb.K, fb.camera_stride, fb.transl_stride, fb.reserved = int(K), int(camera_stride), int(transl_stride), 0
    return fb


class _FramesBase:
    """K frames (poses and / or cameras) of the SAME Gaussians per call: ONE dispatch per kernel for the K frames (the ``*_frames``
    entry points of include/sings_hip.h).  The reference hands its model 16 frames per call (``SinGS.forward_chunk``,
    sings_hybrid.py:474-569) and then renders them one by one (gs_trainer.py:684-714); a frame-parallel step renders K frames of the
    same canonical Gaussians per rank.  Workspaces are K consecutive single-frame workspaces in one allocation each; the
    per-Gaussian backward sums the K frames' gradients in registers and writes the gradient row once, in frame order -- bit
    for bit what K single-frame calls leave behind when the first writes the buffer and the others add (``accumulate=1``)."""

    def _alloc(self, P, W, H, K, cap):
        self.lib = _lib.load()
        if not 1 <= K <= _lib.MAX_FRAMES:
            raise ValueError(f"K must be 1..{_lib.MAX_FRAMES}")
        self.P, self.W, self.H, self.K, self.cap = int(P), int(W), int(H), int(K), int(cap)
        L = _lib.SgLayout()
        sizes = [C.c_size_t() for _ in range(4)]
        _lib.check(self.lib.sg_frames_layout(self.P, self.W, self.H, self.cap, self.K, C.byref(L), *[C.byref(x) for x in sizes]),
                   "sg_frames_layout")
        self.L = L
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.geom = torch.empty(sizes[0].value, **u8)
        self.binning = torch.empty(sizes[1].value, **u8)
        self.binning.view(self.K, L.bin_bytes)[:, :L.bin_ranges].zero_()     # counters zeroed ONCE (SG_FLAG_WS_CLEAN afterwards)
        self._clean = True
        self.img = torch.empty(sizes[2].value, **u8)
        self.bwd_ws = torch.empty(sizes[3].value, **u8)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.color = torch.empty((self.K, 3, self.H, self.W), **f32)
        self.radii = torch.empty((self.K, self.P), dtype=torch.int32, device=self.dev)
        self.d_means2D = torch.empty((self.K, self.P, 3), **f32)
        self._keep, self._s, self._fb, self._cam_stride = [], None, None, 0
        self.throughput = True            # K frames share the chip: no latency-only work (SG_FLAG_THROUGHPUT), as in ViewBatch
        self._chain = None                # inside ViewBatch.run (a batch of frames is one "view" of the ViewBatch): see RasterEngine

    def _chain_state(self, accumulate):
        """(accumulate, done event) for this backward: inside a ViewBatch the batch adds to its gradient row iff an earlier batch
        of the step wrote it (and, with one row across streams, waits for that batch's per-Gaussian half first)."""
        acc, after, done = self._chain() if self._chain is not None else (False, None, None)
        if after is not None:
            torch.cuda.current_stream(self.dev).wait_event(after)
        return bool(accumulate) or bool(acc), done

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def set_camera(self, raster_settings, long_rows=False):
        """One camera for all K frames ([4,4] matrices) or one per frame (``viewmatrix`` / ``projmatrix`` [K,4,4], ``campos``
        [K,3]: the other fields are shared).  ``long_rows``: SG_FLAG_LONG_ROWS (see SkinnedEngine.set_camera)."""
        self._rows = _lib.FLAG_LONG_ROWS if long_rows else 0
        self._keep = []
        vm = raster_settings.viewmatrix
        per_frame = vm.dim() == 3
        if per_frame and (vm.shape[0] != self.K or raster_settings.projmatrix.shape[0] != self.K or raster_settings.campos.shape[0] != self.K):
            raise ValueError(f"per-frame cameras must be stacked [{self.K},4,4] / [{self.K},3]")
        self._s = _settings_struct(raster_settings, self.dev, self.M, self._keep)
        self._cam_stride = 1 if per_frame else 0

    def num_rendered(self):
        nr = (C.c_int64 * self.K)()
        _lib.check(self.lib.sg_read_num_rendered_frames(_ptr(self.binning), self.P, self.W, self.H, self.cap, self.K, nr,
                                                        self._stream()), "read R")
        return [int(v) for v in nr]

    def _flags(self):
        f = ((_lib.FLAG_WS_CLEAN if self._clean else 0) | (_lib.FLAG_THROUGHPUT if self.throughput else 0) |
             (_lib.FLAG_SH_PLANAR if getattr(self, "sh_planar", False) else 0) | getattr(self, "_rows", 0))
        return f

    active_floats = _active_floats


class SkinnedFramesEngine(_FramesBase):
    """LBS-fused path, K posed frames of the same canonical Gaussians per call.  Flat gradient layout as ``SkinnedEngine``
    (xyz_canon 3P, scales 3P, opacity P, sh 3MP, [rot_canon]); per frame: ``color`` [K,3,H,W], ``radii`` [K,P], ``d_means2D``
    [K,P,3], ``d_A`` [K,J,16], ``d_transl`` [K,3]."""

    def __init__(self, P, J, W, H, sh_coeffs, K, device, capacity_pairs, with_rot=False, grad_flat=None, rot_width=9, sh_planar=False):
        self.dev = torch.device(device)
        self.J, self.M = int(J), int(sh_coeffs)
        self.sh_planar = bool(sh_planar)
        self._alloc(P, W, H, K, capacity_pairs)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.skin_ws = torch.empty(int(self.lib.sg_skin_ws_floats_frames(self.P, self.K)), **f32)
        rot_width = int(rot_width)
        per = 3 + 3 + 1 + 3 * self.M + (rot_width if with_rot else 0)
        if grad_flat is not None and (grad_flat.numel() != self.P * per or grad_flat.dtype != torch.float32
                                      or not grad_flat.is_contiguous() or grad_flat.device != self.dev):
            raise ValueError(f"grad_flat must be a contiguous fp32 tensor of {self.P * per} elements on {self.dev}")
        self.grad_flat = torch.empty(self.P * per, **f32) if grad_flat is None else grad_flat.view(-1)
        o = 0

        def carve(n, *shape):
            nonlocal o
            v = self.grad_flat[o:o + n].view(*shape); o += n
            return v
        _carve_skinned(self, carve, with_rot, rot_width, grad_flat is None)
        self.d_A = torch.empty((self.K, self.J, 16), **f32); self.d_transl = torch.empty((self.K, 3), **f32)
        self._k = None

    def rebind_gradients(self, grad_flat):
        """Point the gradient views at another flat buffer (several engines of one step sharing ONE buffer: FramePipeline)."""
        if grad_flat.numel() != self.grad_flat.numel() or grad_flat.dtype != torch.float32 or not grad_flat.is_contiguous():
            raise ValueError("rebind_gradients: a contiguous fp32 buffer of the same size")
        with_rot, rw = self.d_rot is not None, (self.d_rot.shape[1] if self.d_rot is not None else 0)
        self.grad_flat = grad_flat.view(-1)
        o = 0

        def carve(n, *shape):
            nonlocal o
            v = self.grad_flat[o:o + n].view(*shape); o += n
            return v
        _carve_skinned(self, carve, with_rot, rw, False)

    def set_frames(self, xyz_canon, rotmat_canon, lbs_weights, A, smpl_scale, transl):
        """``A``: [K,J,4,4] / [K,J,16] joint transforms of the K frames; ``transl``: [K,3] (one per frame), [3] (shared) or None."""
        from .skinned import _skin_struct
        A = A.reshape(self.K, -1, 16)
        if A.shape[1] != self.J:
            raise ValueError(f"A must be [{self.K},{self.J},16]")
        self._kkeep = []
        tstride = 0
        if transl is not None:
            transl = transl.contiguous().float()
            if transl.numel() == 3:
                tstride = 0
            elif transl.numel() == 3 * self.K:
                tstride = 3
            else:
                raise ValueError(f"transl must be [{self.K},3] or [3]")
        # (_skin_struct reads J from A's row count: hand it frame 0, the pointer is the base of the K stacked frames)
        self._k = _skin_struct(self.dev, xyz_canon, rotmat_canon, lbs_weights, A[0], smpl_scale,
                               None if transl is None else transl.reshape(-1)[:3], None, self._kkeep)
        A = A.contiguous()
        self._kkeep += [A, transl]
        self._k.A = A.data_ptr()
        self._k.transl = None if transl is None else transl.data_ptr()
        self._fb = _frame_batch(self.K, self._cam_stride, tstride)

    def forward(self, shs, opacities, scales, sync_num_rendered=False, phase=None, posed_out=None):
        """``phase``: None = the whole forward; "binning" / "composite" = its two halves as separate calls on the same workspaces
        (SG_FLAG_FORWARD_BINNING / _COMPOSITE), for a caller that runs them on different streams and orders them with events.
        ``posed_out``: optional (xyz [K,P,3], quaternions [K,P,4], scales [K,P,3]) fp32 tensors the posed values of the K frames are
        also written to (what forward_chunk returns, sings_hybrid.py:512-553; the render itself never reads them back)."""
        if self._s is None or self._fb is None or self._k is None:
            raise RuntimeError("SkinnedFramesEngine: set_camera() and set_frames() first")
        if posed_out is not None:
            for x, wdt in zip(posed_out, (3, 4, 3)):
                if (tuple(x.shape) != (self.K, self.P, wdt) or x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.dev):
                    raise ValueError(f"posed_out: contiguous fp32 [{self.K},{self.P},3|4|3] tensors on {self.dev}")
        pxyz, pq, psc = posed_out if posed_out is not None else (None, None, None)
        nr = (C.c_int64 * self.K)()
        self._fb.camera_stride = self._cam_stride
        self._s.flags = self._flags() | {None: 0, "binning": _lib.FLAG_FORWARD_BINNING, "composite": _lib.FLAG_FORWARD_COMPOSITE}[phase]
        if phase == "composite":
            self._s.flags &= ~_lib.FLAG_WS_CLEAN
        self._clean = False
        _lib.check(self.lib.sg_skinned_forward_frames(
            C.byref(self._s), C.byref(self._fb), self.P, C.byref(self._k), _ptr(shs), _ptr(opacities), _ptr(scales),
            _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.color), _ptr(self.radii), _ptr(pxyz), _ptr(pq),
            _ptr(psc), nr if sync_num_rendered and phase != "composite" else None, self._stream()), "skinned forward (frames)")
        self._clean = phase != "binning"                      # (the COMPOSITE leaves the counters zeroed for the next forward)
        return [int(v) for v in nr] if sync_num_rendered and phase != "composite" else None

    def backward(self, shs, opacities, scales, dL_dcolor, accumulate=False):
        """``dL_dcolor`` [K,3,H,W].  ``accumulate``: the K frames' sum is ADDED to the gradient buffer (a later batch of the step)."""
        self.backward_records(dL_dcolor)
        self.backward_gaussians(shs, opacities, scales, accumulate)

    def backward_records(self, dL_dcolor):
        """First half of ``backward``: the per-tile composite backward of the K frames (VALU-bound)."""
        _lib.check(self.lib.sg_rasterize_backward_records_frames(
            C.byref(self._s), C.byref(self._fb), self.P, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img),
            _ptr(self.bwd_ws), _ptr(dL_dcolor), self._stream()), "skinned backward (records, frames)")

    def backward_gaussians(self, shs, opacities, scales, accumulate=False):
        """Second half: record sums + the per-Gaussian chain rule summed over the K frames (latency-bound)."""
        accumulate, done = self._chain_state(accumulate)
        _lib.check(self.lib.sg_skinned_backward_gaussians_frames(
            C.byref(self._s), C.byref(self._fb), self.P, C.byref(self._k), _ptr(shs), _ptr(opacities), _ptr(scales),
            _ptr(self.radii), _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.bwd_ws), _ptr(self.skin_ws),
            int(bool(accumulate)), None, None, _ptr(self.d_xyz), _ptr(self.d_rot), _ptr(self.d_scales), _ptr(self.d_opacity),
            _ptr(self.d_sh), _ptr(self.d_means2D), _ptr(self.d_A), _ptr(self.d_transl), self._stream()),
            "skinned backward (gaussians, frames)")
        if done is not None:
            done.record(torch.cuda.current_stream(self.dev))


class RasterFramesEngine(_FramesBase):
    """Un-skinned path: K cameras of the same Gaussians per call (``sg_rasterize_*_frames``).  Flat gradient layout as
    ``RasterEngine``; ``color`` [K,3,H,W], ``radii`` [K,P], ``d_means2D`` [K,P,3]."""

    def __init__(self, P, W, H, sh_coeffs, K, device, capacity_pairs, grad_flat=None, sh_planar=False):
        self.dev = torch.device(device)
        self.M = int(sh_coeffs)
        self.sh_planar = bool(sh_planar)        # dL/dsh coefficient-major [M,P,3] (the SH block is LAST in this layout: the planes in
        self._alloc(P, W, H, K, capacity_pairs)  # use are a prefix of it, active_floats(deg) floats of grad_flat carry gradient)
        f32 = dict(dtype=torch.float32, device=self.dev)
        per = 3 + 3 + 4 + 1 + 3 * self.M
        if grad_flat is not None and (grad_flat.numel() != self.P * per or grad_flat.dtype != torch.float32
                                      or not grad_flat.is_contiguous() or grad_flat.device != self.dev):
            raise ValueError(f"grad_flat must be a contiguous fp32 tensor of {self.P * per} elements on {self.dev}")
        self.grad_flat = torch.empty(self.P * per, **f32) if grad_flat is None else grad_flat.view(-1)
        o = 0

        def carve(n, *shape):
            nonlocal o
            v = self.grad_flat[o:o + n].view(*shape); o += n
            return v
        self.d_means3D = carve(self.P * 3, self.P, 3); self.d_scales = carve(self.P * 3, self.P, 3)
        self.d_rots = carve(self.P * 4, self.P, 4); self.d_opacity = carve(self.P, self.P, 1)
        self._head = o
        if not self.M:
            self.d_sh = None
        elif self.sh_planar:
            self.d_sh = carve(self.P * 3 * self.M, self.M, self.P, 3)
        else:
            self.d_sh = carve(self.P * 3 * self.M, self.P, self.M, 3)
        self._hint = 0

    def set_camera(self, raster_settings, short_lists=False, long_rows=False):
        super().set_camera(raster_settings, long_rows=long_rows)
        self._hint = _lib.FLAG_SHORT_LISTS if short_lists else 0
        self._fb = _frame_batch(self.K, self._cam_stride, 0)

    def forward(self, means3D, shs, opacities, scales, rotations, sync_num_rendered=False):
        if self._s is None or self._fb is None:
            raise RuntimeError("RasterFramesEngine: set_camera() first")
        nr = (C.c_int64 * self.K)()
        self._s.flags = self._hint | self._flags()
        self._clean = False
        _lib.check(self.lib.sg_rasterize_forward_frames(
            C.byref(self._s), C.byref(self._fb), self.P, _ptr(means3D), _ptr(shs), None, _ptr(opacities), _ptr(scales),
            _ptr(rotations), None, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img), _ptr(self.color),
            _ptr(self.radii), nr if sync_num_rendered else None, self._stream()), "forward (frames)")
        self._clean = True
        return [int(v) for v in nr] if sync_num_rendered else None

    def backward(self, means3D, shs, opacities, scales, rotations, dL_dcolor, accumulate=False):
        _lib.check(self.lib.sg_rasterize_backward_records_frames(
            C.byref(self._s), C.byref(self._fb), self.P, _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.img),
            _ptr(self.bwd_ws), _ptr(dL_dcolor), self._stream()), "backward (records, frames)")
        accumulate, done = self._chain_state(accumulate)
        _lib.check(self.lib.sg_rasterize_backward_gaussians_frames(
            C.byref(self._s), C.byref(self._fb), self.P, _ptr(means3D), _ptr(shs), None, _ptr(opacities), _ptr(scales),
            _ptr(rotations), None, _ptr(self.radii), _ptr(self.geom), _ptr(self.binning), self.cap, _ptr(self.bwd_ws),
            int(bool(accumulate)), _ptr(self.d_means3D), _ptr(self.d_means2D), _ptr(self.d_sh), None, _ptr(self.d_opacity),
            _ptr(self.d_scales), _ptr(self.d_rots), None, self._stream()), "backward (gaussians, frames)")
        if done is not None:
            done.record(torch.cuda.current_stream(self.dev))


class FramePipeline:
    """The batches of frames of ONE optimisation step on TWO streams, by kind of kernel instead of by batch.

    A frame's chain alternates between kernels that keep the vector ALUs busy (the two composite kernels: half of an avatar frame)
    and kernels that wait -- binning, long-list sort, loss, record sums, per-Gaussian backward: a handful of waves per SIMD in
    dependent chains, a fifth of the chip's issue slots.  Batch-per-stream scheduling (``ViewBatch``) overlaps them only by
    accident.  Here stream A runs nothing but the composite kernels of all batches, back to back; stream B (high priority, so that
    its short kernels get wave slots as soon as they are ready) runs everything else, issued in an order that always has the
    other batch's light work ready beside the running composite:

        B:  bin 0 | bin 1 | .. | loss 0 | loss 1 | .. | per-Gaussian 0 | per-Gaussian 1 | ..
        A:          fwd 0 | fwd 1 | ..   | records 0 | records 1 | ..

    (events order a batch's own phases across the streams).  The per-Gaussian halves run in batch order on B and share ONE gradient
    buffer -- batch 0 writes, the others add: bit for bit the sum that the same batches give one after the other on one stream.
    Engines: ``SkinnedFramesEngine`` objects bound to the same ``grad_flat`` (``rebind_gradients``)."""

    def __init__(self, engines, device):
        self.engines = list(engines)
        self.dev = torch.device(device)
        self.A = torch.cuda.Stream(self.dev)
        self.B = torch.cuda.Stream(self.dev, priority=-1)
        self.ev = [[torch.cuda.Event() for _ in range(4)] for _ in self.engines]
        for e in self.engines:
            e.throughput = True
            if e.grad_flat.data_ptr() != self.engines[0].grad_flat.data_ptr():
                raise ValueError("FramePipeline: every engine must be bound to the same gradient buffer (rebind_gradients)")

    def run(self, prepare, forward_args, loss):
        """``prepare(b, engine)``: set the batch's frames (runs on B); ``forward_args`` = (shs, opacities, scales);
        ``loss(b, engine)`` -> dL_dcolor [K,3,H,W] (runs on B, after the batch's composite)."""
        cur = torch.cuda.current_stream(self.dev)
        A, B, ev = self.A, self.B, self.ev
        A.wait_stream(cur); B.wait_stream(cur)
        for b, e in enumerate(self.engines):
            with torch.cuda.stream(B):
                prepare(b, e)
                e.forward(*forward_args, phase="binning")
                ev[b][0].record(B)
            with torch.cuda.stream(A):
                A.wait_event(ev[b][0])
                e.forward(*forward_args, phase="composite")
                ev[b][1].record(A)
        grads = []
        for b, e in enumerate(self.engines):
            with torch.cuda.stream(B):
                B.wait_event(ev[b][1])
                grads.append(loss(b, e))
                ev[b][2].record(B)
            with torch.cuda.stream(A):
                A.wait_event(ev[b][2])
                e.backward_records(grads[b])
                ev[b][3].record(A)
        for b, e in enumerate(self.engines):
            with torch.cuda.stream(B):
                B.wait_event(ev[b][3])
                e.backward_gaussians(*forward_args, accumulate=b > 0)
        cur.wait_stream(A); cur.wait_stream(B)
