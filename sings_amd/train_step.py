"""One SinGS training step on one posed frame, composed from the library's pieces (what ``scripts/train_avatar.py`` ->
``GaussianTrainer.train`` does per iteration, SURVEY.md 3.1, without the optimiser / densification bookkeeping):

    decode_attributes (tri-plane + decoders)          sings_hybrid.py:249-313      sings_amd/decode.py
    fused LBS + projection + binning + composite      sings_hybrid.py:398-428, gs_renderer_single.py:12-107   sings_amd/skinned.py
    clamp + L1 + SSIM                                 losses/loss.py:55-69         sings_amd/photo_loss.py
    regularisers (optional)                           gs_trainer.py:355-399        sings_amd/regularizers.py
    backward through all of it

Everything between the parameters and the scalar loss runs in HIP kernels (library GEMMs for the decoders' forward /
input-gradient products); torch autograd only strings the stages together.  The regularisers depend on the decoded
attributes only, not on the render: with ``overlap_regularisers`` (default) they run on a second HIP stream next to the
LBS + raster + photometric-loss chain, also inside a captured HIP graph.  WHERE they run decides what they cost
(profiles/r03_train_step_trace.log): the k-NN query keeps 9 waves per SIMD resident on every CU for 0.23 ms, and beside it the
raster forward's latency-bound chain took 2-7x as long per kernel (the long-list sort needs a nearly empty CU per workgroup:
146 us instead of 21).  With ``defer_regulariser_join`` the query is held until the raster forward has finished and is NOT
joined before the loss: ``forward`` returns the photometric and the regulariser loss as two roots (``extras["loss_roots"]``) and
``AvatarStep.backward`` STAGES the backward pass by hand (round 5): the render and the regularisers read detached views of the
decoded attributes; the photometric gradients (loss, composite, LBS: this stream) are one ``torch.autograd.grad`` call, L2Norm's
gradients -- computed by its kernel with the value, in front of the query on the side stream -- are added with one ``_foreach_add_``,
and the decoders' backward starts from the sums.  The k-NN regulariser's gradient (scales only) enters at an identity node placed on
the geometry decoder's outputs BEFORE the appearance decoder is issued: autograd visits the youngest nodes first, so the appearance
decoder's backward is issued -- and runs beside the query -- before the main stream waits for the side stream at that node.  The
order of the launches is then the order the graph executor needs to keep the critical chain on ONE queue (sings_amd/decode.py
"capture order"; 2.18 -> 1.97-2.05 ms per step with the deferred weight gradients, profiles/r05_graph_queues.log).
"""
import torch

from . import decode as _dec
from .decode import decode_attributes
from .photo_loss import photometric_loss, photometric_loss_frames
from .skinned import rasterize_skinned_frames, rasterize_skinned_gaussians


class _AddScalars(torch.autograd.Function):
    """a + b of two 0-dim losses: ONE launch forward and none backward (torch.stack(...).sum() is a concatenation and a reduction
    forward, and autograd's sum / stack backward each way; between the loss kernels and the first backward kernel every such
    4-us launch sits on the step's critical path)."""

    @staticmethod
    def forward(ctx, a, b):
        return a + b

    @staticmethod
    def backward(ctx, g):
        return g, g


class _SumFrames(torch.autograd.Function):
    """(l1[K].sum(), ssim[K].sum()) in one node whose backward hands BOTH per-frame terms the SAME upstream tensor when the two sums got
    the same one (they do under ``_AddScalars``): the photometric loss then returns the gradient it computed in the forward, times
    that tensor, instead of marching over the K images again (sings_amd.photo_loss._PhotoLossFrames.backward)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.k = int(a.shape[0])
        return a.sum(), b.sum()

    @staticmethod
    def backward(ctx, ga, gb):
        ea = ga.expand(ctx.k)
        return ea, (ea if gb is ga else gb.expand(ctx.k))


class _Inject(torch.autograd.Function):
    """Identity in the forward; the backward ADDS a gradient that arrives late from another stream (``slot[key]``: a tensor or
    None, ``slot["ready"]``: the event behind the kernel that produces it).  Placed on the geometry decoder's outputs before the appearance decoder is
    issued, so autograd reaches it -- and waits for that stream -- only after the appearance decoder's backward has been issued."""

    @staticmethod
    def forward(ctx, x, slot, key):
        ctx.slot, ctx.key = slot, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        extra = ctx.slot.pop(ctx.key, None)
        if extra is not None:
            cur = torch.cuda.current_stream(g.device)
            cur.wait_event(ctx.slot["ready"])                    # (not the whole side stream: more may have been issued behind it)
            extra.record_stream(cur)
            g = g + extra
        return g, None, None


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


def capture_step(fn, warmup=3, device=None):
    """``graph, outputs = capture_step(fn)``: record ``fn()`` -- one whole step: forward, backward, whatever it launches -- into
    a HIP graph the safe way.  On ROCm 7.2 ``hipStreamEndCapture`` SEGFAULTS (not an error code: the process dies) when
    something ``fn`` returned still carries its autograd graph at the end of the capture (gpurun_out/crash.log, round 3): the
    saved tensors of that graph live in the capture's private memory pool on several streams.  This wrapper therefore looks at
    what ``fn`` returned BEFORE it ends the capture: if any returned tensor requires grad / has a grad_fn, the references are
    dropped first, the capture is ended cleanly (the graph is discarded) and a RuntimeError says what to do -- return
    ``loss.detach()`` (or ``.detach().clone()``) from the captured callable.  ``warmup`` eager runs on a side stream come first
    (allocator warm-up, lazily created streams), as ``torch.cuda.graph`` requires."""
    import gc
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("capture_step: a capture is already in progress on this stream")
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(max(0, int(warmup))):
            out = fn()
            n_bad = sum(1 for t_ in _tensors_of(out) if t_.requires_grad or t_.grad_fn is not None)
            del out
            if n_bad:
                torch.cuda.current_stream(dev).wait_stream(side)
                raise RuntimeError(f"capture_step: the callable returns {n_bad} tensor(s) that still carry an autograd graph; "
                                   "return detached values (loss.detach()) -- ending a HIP-graph capture with them alive "
                                   "crashes hipStreamEndCapture on ROCm 7.2")
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    n_bad = 0
    with torch.cuda.graph(graph):
        out = fn()
        n_bad = sum(1 for t_ in _tensors_of(out) if t_.requires_grad or t_.grad_fn is not None)
        if n_bad:                              # (only reachable with warmup = 0) drop every reference BEFORE the capture ends
            out = None
            gc.collect()
    if n_bad:
        del graph
        raise RuntimeError(f"capture_step: the callable returned {n_bad} tensor(s) that still carry an autograd graph; the capture "
                           "was discarded.  Return detached values (loss.detach())")
    return graph, out


class AvatarStep(torch.nn.Module):
    def __init__(self, xyz_anchor, lbs_weights, triplane, geometry_dec, appearance_dec, l1_w=0.8, ssim_w=0.2,
                 thickness_factor=1.0, scaling_multiplier=None, l2_norm=None, gaussian_connect=None, gaussian_connect_w=0.0,
                 overlap_regularisers=True, defer_regulariser_join=False):
        super().__init__()
        self.overlap_regularisers, self._side = bool(overlap_regularisers), None
        self.defer_regulariser_join = bool(defer_regulariser_join)
        self.xyz = torch.nn.Parameter(xyz_anchor.detach().clone())
        self.register_buffer("lbs_weights", lbs_weights.detach().clone())
        self.triplane, self.geometry_dec, self.appearance_dec = triplane, geometry_dec, appearance_dec
        self.l1_w, self.ssim_w, self.thickness_factor = l1_w, ssim_w, thickness_factor
        self.scaling_multiplier = scaling_multiplier
        self.l2_norm, self.gaussian_connect, self.gaussian_connect_w = l2_norm, gaussian_connect, gaussian_connect_w
        # what was captured / sized for this set of Gaussians (sings_amd.train_loop): moves on with every densify / prune / SH-degree change
        self.topology_version = 0
        self.xyz_gradient_accum = self.denom = self.max_radii2D = None

    # ---- densification statistics and topology (sings_hybrid.py:1013-1015, 930-932; gs_trainer.py:486-492)
    def enable_densification_stats(self):
        """(Re)allocate the statistics for the current set: from now on ``forward`` hands the rasterizer a ``means2D`` holder and
        ``add_densification_stats(extras)`` after the backward accumulates |screen-space gradient|, the visibility count and the
        largest radius per Gaussian."""
        n, dev = int(self.xyz.shape[0]), self.xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.denom = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.max_radii2D = torch.zeros(n, dtype=torch.float32, device=dev)

    def add_densification_stats(self, extras):
        """``xyz_gradient_accum[vis] += |viewspace_points.grad[vis, :2]|``, ``denom[vis] += 1``, ``max_radii2D[vis] = max(., radii[vis])``
        for every frame of the step.  Masked arithmetic instead of boolean indexing: the same sums without a host round trip, so the
        call may sit inside a captured step."""
        if self.xyz_gradient_accum is None:
            return
        g, radii = extras["viewspace_points"].grad, extras["radii"]
        if g is None:
            raise RuntimeError("add_densification_stats: call it after the backward pass of the step (viewspace_points.grad is empty)")
        g, radii = g.reshape(-1, g.shape[-2], 3), radii.reshape(-1, radii.shape[-1])
        vis = (radii > 0).to(torch.float32)                                       # [K, N]
        self.xyz_gradient_accum.add_((torch.norm(g[..., :2], dim=-1) * vis).sum(0).unsqueeze(1))
        self.denom.add_(vis.sum(0).unsqueeze(1))
        torch.maximum(self.max_radii2D, radii.to(torch.float32).max(0).values, out=self.max_radii2D)

    def set_topology(self, xyz_anchor, lbs_weights, scaling_multiplier=None):
        """A new set of Gaussians (densify / prune): new anchor PARAMETER (the old object is dead: an optimiser, a gradient arena or a
        captured graph that still holds it must be rebuilt -- ``topology_version`` says so), new skinning weights, statistics reset."""
        if xyz_anchor.shape[0] != lbs_weights.shape[0] or xyz_anchor.dim() != 2 or xyz_anchor.shape[1] != 3:
            raise ValueError("set_topology: xyz_anchor [N,3] and lbs_weights [N,J] must describe the same N Gaussians")
        self.xyz = torch.nn.Parameter(xyz_anchor.detach().clone().contiguous())
        self.lbs_weights = lbs_weights.detach().clone().contiguous()
        if scaling_multiplier is not None:
            self.scaling_multiplier = scaling_multiplier
        elif torch.is_tensor(self.scaling_multiplier) and self.scaling_multiplier.dim() > 0 and self.scaling_multiplier.shape[0] != xyz_anchor.shape[0]:
            raise ValueError("set_topology: the per-Gaussian scaling_multiplier must be passed for the new set")
        self.topology_version += 1
        _dec.invalidate_gradient_arena()
        if self.xyz_gradient_accum is not None:
            self.enable_densification_stats()

    def forward(self, A_cano2pose, raster_settings, gt_rgb, mask, bg_color, smpl_scale=None, transl=None):
        """-> (loss, loss_dict, extras).  ``A_cano2pose`` [J,4,4]: one frame per step (the reference's step); [K,J,4,4]: a chunk
        of K <= 16 frames of the step's Gaussians -- ONE attribute decode, the K frames rendered, compared (``gt_rgb`` / ``mask``
        [K,...] or one for all) and differentiated in one call per direction, the photometric terms summed over the frames,
        the regularisers once (``transl`` [K,3] / [3]; cameras: one, or stacked [K,4,4] / [K,3] in ``raster_settings``).  With ``defer_regulariser_join`` (and gradients enabled, regularisers present) ``loss``
        is **None**: the photometric and the regulariser terms are two autograd roots (``extras["loss_roots"]``) -- call
        ``self.backward(loss_dict, extras)``, which runs both, joins the regularisers' side stream and fills
        ``loss_dict["loss"]``.  Until then the regulariser entries of ``loss_dict`` live on that UNJOINED stream: do not read
        them (``.item()``, logging) before ``backward`` -- or call ``self.join_regularisers(loss_dict)`` first.
        Inside a HIP-graph capture: use ``sings_amd.train_step.capture_step`` (it refuses, with an error instead of a dead
        process, a captured callable that returns tensors still carrying their autograd graph), and keep the regularisers on
        the ONE side stream this module owns (``GaussiansEdgeLoss.finish`` refuses any other stream while capturing)."""
        has_reg = self.l2_norm is not None or (self.gaussian_connect is not None and self.gaussian_connect_w > 0)
        has_knn = self.gaussian_connect is not None and self.gaussian_connect_w > 0
        overlap = has_reg and self.overlap_regularisers and self.xyz.is_cuda
        defer = (overlap and self.defer_regulariser_join and torch.is_grad_enabled() and has_knn
                 and hasattr(self.gaussian_connect, "prepare"))
        inject = {} if defer else None                           # the k-NN regulariser's gradient (scales only: its edge lengths are
        hook = (lambda x, sc: (x, _Inject.apply(sc, inject, "scales"))) if defer else None          # detached, loss_items.py:75)
        # the tri-plane backward's point sorts (decode.py: "the point-only half ... early") are held back: beside the first decoder
        # layers they cost the forward chain ~70 us (the first layer: 121 instead of 53 us); here they are issued on the regularisers'
        # stream behind the k-NN query, long before the tri-plane backward needs them
        late_prev = _dec._TP["late"]
        _dec._TP["late"] = bool(defer)
        del _dec._TP_PENDING[:]                                  # (a preparation left behind by a forward that raised)
        try:
            attrs = decode_attributes(self.xyz, self.triplane, self.geometry_dec, self.appearance_dec, self.thickness_factor,
                                      self.scaling_multiplier, geometry_hook=hook)
        finally:
            _dec._TP["late"] = late_prev
        # Two roots (defer): the render and the regularisers read DETACHED views of the decoded attributes (`use`), so that the
        # backward pass can be staged by hand (AvatarStep.backward): photometric gradients on this stream, the regularisers' on the
        # side stream, one addition, then the decoders' backward -- in an order the graph executor turns into ONE queue for the
        # critical chain (sings_amd/decode.py "capture order").  One root: `use` IS `attrs`, plain autograd.
        use = attrs
        if defer:
            use = {k: (v.detach().requires_grad_(v.requires_grad) if torch.is_tensor(v) else v) for k, v in attrs.items()}
        # anisotropic: the decoder's 6-D rotations go to the fused kernels as they are (rotation_6d_to_matrix of
        # sings_hybrid.py:356-357 runs inside them); isotropic: None
        rot = use["rot6d_canon"]
        reg = {}

        def l2():
            if self.l2_norm is not None:
                reg["l2"] = self.l2_norm({"xyz_offsets": use["xyz_offsets"], "scales": use["scales"], "opacity": use["opacity"]})

        def regularisers():
            l2()
            if has_knn:
                reg["gaussian_connect_loss"] = self.gaussian_connect_w * self.gaussian_connect(
                    {"xyz_canon": use["xyz_canon"], "scales": use["scales"]})

        side = None
        if overlap:
            dev = attrs["xyz_canon"].device
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(dev)
            side, cur = self._side, torch.cuda.current_stream(dev)
            if defer:
                decoded = torch.cuda.Event()
                decoded.record(cur)                              # fork point; the side stream's work is ISSUED after the photometric loss
            else:
                side.wait_stream(cur)                            # fork: the decoded attributes are complete
                with torch.cuda.stream(side):
                    regularisers()
        frames = A_cano2pose.dim() == 4                          # [K,J,4,4]: a CHUNK of K frames of this step's Gaussians (round 4)
        # the screen-space gradient the densifier reads (gs_renderer_single.py:50-56): a holder per step, only when statistics are kept
        m2d = None
        if self.xyz_gradient_accum is not None and torch.is_grad_enabled():
            shape = (int(A_cano2pose.shape[0]), int(self.xyz.shape[0]), 3) if frames else (int(self.xyz.shape[0]), 3)
            m2d = torch.zeros(shape, dtype=torch.float32, device=self.xyz.device, requires_grad=True)
        if frames:
            # one decode, K frames rendered and differentiated in one call per direction: the attribute decode -- 3/4 of a
            # one-frame step -- is paid once per step, not once per frame (gt_rgb / mask: [K,...] or one for all frames)
            color, radii = rasterize_skinned_frames(use["xyz_canon"], rot, use["scales"], use["opacity"], use["shs"],
                                                    self.lbs_weights, A_cano2pose, raster_settings, smpl_scale=smpl_scale,
                                                    transl=transl, means2D=m2d)
        else:
            color, radii = rasterize_skinned_gaussians(use["xyz_canon"], rot, use["scales"], use["opacity"], use["shs"],
                                                       self.lbs_weights, A_cano2pose, raster_settings, smpl_scale=smpl_scale,
                                                       transl=transl, means2D=m2d)
        if defer:
            rendered = torch.cuda.Event()
            rendered.record(cur)
        if frames:
            per_frame, extras = photometric_loss_frames(color, gt_rgb, mask, bg_color, self.l1_w, self.ssim_w)
            if len(per_frame) == 2:                                         # the step's photometric terms: summed over its frames
                loss_dict = dict(zip(per_frame.keys(), _SumFrames.apply(*per_frame.values())))
            else:
                loss_dict = {k: v.sum() for k, v in per_frame.items()}
            extras = dict(extras, per_frame=per_frame)
        else:
            loss_dict, extras = photometric_loss(color, gt_rgb, mask, bg_color, self.l1_w, self.ssim_w)
        if defer:
            # two roots, no join: the regularisers' gradients are waited for where they are consumed (AvatarStep.backward)
            vals = [v.reshape(()) for v in loss_dict.values()]
            photo_root = _AddScalars.apply(vals[0], vals[1]) if len(vals) == 2 else torch.stack(vals).sum()
            # the regularisers, ISSUED last (the main chain keeps its queue in the graph executor, sings_amd/decode.py "capture
            # order") but waiting only for what they read: L2Norm and the k-NN grids for the decode, the k-NN QUERY -- 9 waves per
            # SIMD on every CU for 0.23 ms -- for the raster forward.  (Measured, profiles/r05_graph_queues.log: the executor starts
            # a second queue's chain 60-160 us after its dependency is met, so part of the query is still exposed behind the
            # backward composite; issued FIRST, as in round 4, the chain starts at once and the raster forward is what waits.)
            # No autograd on the side stream: both regularisers' kernels compute their gradients with the value (`last_grads`), and the
            # engine's stream bookkeeping around an autograd call put a wait for THIS stream's newest kernel in front of them.
            side.wait_event(decoded)
            l2_grads = None
            det = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in use.items()}
            with torch.cuda.stream(side):
                if self.l2_norm is not None:
                    # in front of the k-NN query on this stream: the appearance decoder's backward needs the opacity term
                    reg["l2"] = self.l2_norm({"xyz_offsets": det["xyz_offsets"], "scales": det["scales"], "opacity": det["opacity"]})
                    l2_grads = (dict(self.l2_norm.last_grads), torch.cuda.Event())
                    l2_grads[1].record(side)
                edge = self.gaussian_connect.prepare({"xyz_canon": det["xyz_canon"], "scales": det["scales"]})
                side.wait_event(rendered)
                try:
                    self.gaussian_connect.finish()
                except BaseException:
                    self.gaussian_connect.abort()
                    raise
                g_edge = self.gaussian_connect.last_grads["scales"]
                if self.gaussian_connect_w != 1.0:
                    edge, g_edge = self.gaussian_connect_w * edge, self.gaussian_connect_w * g_edge
                reg["gaussian_connect_loss"] = edge
                if use["scales"].requires_grad:
                    inject["scales"] = g_edge                    # reaches the decoders' backward through the _Inject node
                reg_root = edge.reshape(())
                inject["ready"] = torch.cuda.Event()
                inject["ready"].record(side)
                _dec.flush_triplane_prepare(side)                # (same stream, behind the query: executed in this order)
            loss_dict.update(reg)
            extras["loss_roots"] = (photo_root, reg_root)
            extras["staged"] = (attrs, use, inject, l2_grads)
            return None, loss_dict, {"render_raw": color, "radii": radii, "attrs": attrs, "viewspace_points": m2d, **extras}
        if side is not None:
            cur.wait_stream(side)                                # join before the loss terms meet
            for v in reg.values():
                v.record_stream(cur)
        elif has_reg:
            regularisers()
        # l1 + ssim first, under ONE node that hands both the same upstream tensor (the loss kernel's backward is then a multiplication:
        # sings_amd.photo_loss), the regularisers added behind
        vals = [v.reshape(()) for v in loss_dict.values()]
        loss = _AddScalars.apply(vals[0], vals[1]) if len(vals) >= 2 else vals[0]
        for v in vals[2:]:
            loss = loss + v
        for v in reg.values():
            loss = loss + v.reshape(())
        loss_dict.update(reg)
        loss_dict["loss"] = loss
        return loss, loss_dict, {"render_raw": color, "radii": radii, "attrs": attrs, "viewspace_points": m2d, **extras}

    def join_regularisers(self, loss_dict=None):
        """Make the current stream wait for the regularisers' side stream (a ``defer_regulariser_join`` forward leaves it
        unjoined), so that the regulariser losses can be read before ``backward``."""
        if self._side is not None:
            cur = torch.cuda.current_stream(self._side.device)
            cur.wait_stream(self._side)
            for v in (loss_dict or {}).values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)

    def backward(self, loss_dict, extras):
        """Backward pass of a ``defer_regulariser_join`` forward, staged by hand (module docstring); then the streams join and
        ``loss_dict["loss"]`` = the sum of the terms.  (A forward that returned a loss: ``loss.backward()`` as usual.)"""
        try:
            return self._backward_staged(loss_dict, extras)
        except BaseException:
            _dec.reset_deferred()                                # (closures with raw gradient pointers must not outlive the pass)
            raise

    def _backward_staged(self, loss_dict, extras):
        photo_root, reg_root = extras["loss_roots"]
        attrs, use, inject, l2_grads = extras["staged"]
        one = getattr(self, "_one", None)
        if one is None or one.device != photo_root.device:
            from .photo_loss import register_unit_scalar
            one = self._one = register_unit_scalar(torch.ones((), dtype=photo_root.dtype, device=photo_root.device))     # (not a fill launch per root and step)
        dev = photo_root.device
        cur, side = torch.cuda.current_stream(dev), self._side
        names = [k for k, v in use.items() if torch.is_tensor(v) and v.requires_grad]
        leaves = [use[k] for k in names]
        m2d = extras.get("viewspace_points")
        # 1. the photometric gradients of the decoded attributes: loss, composite, LBS -- on this stream, issued first
        got = torch.autograd.grad([photo_root], leaves + ([m2d] if m2d is not None else []), [one], allow_unused=True)
        if m2d is not None:
            m2d.grad = got[-1]                                   # (what add_densification_stats reads: the densifier's statistic)
        g = dict(zip(names, got))
        # 2. L2Norm's gradients (computed with its value, in front of the k-NN query on the side stream): ONE addition for the
        #    attributes that have both.  The k-NN regulariser's gradient (scales) is already in `inject`: it reaches the decoders'
        #    backward through the _Inject node, i.e. AFTER the appearance decoder's backward has been issued
        if l2_grads is not None:
            cur.wait_event(l2_grads[1])
            both_a, both_b = [], []
            for k, gr in l2_grads[0].items():
                if gr is None or k not in g:
                    continue
                gr.record_stream(cur)
                if g[k] is None:
                    g[k] = gr
                else:
                    both_a.append(g[k]); both_b.append(gr)
            # (not torch._foreach_add_: its 64 k-element chunks make ten workgroups of the two 150 k-row attributes -- 19 us alone on the
            #  chip, 92 beside the k-NN query of the side stream; an add per tensor is 5 us.  The step itself did not get shorter
            #  for it -- 1.867-1.875 ms either way, same box: its length is the work of BOTH queues, LAB 6.8)
            for x, y in zip(both_a, both_b):
                x.add_(y)
        # 3. the decoders' backward: appearance decoder first (its nodes are the youngest), then the _Inject node waits for the side stream
        roots = [(attrs[k], g[k]) for k in names if g.get(k) is not None]
        torch.autograd.backward([r for r, _ in roots], [x for _, x in roots])
        cur.wait_stream(side)
        for v in list(loss_dict.values()) + [reg_root]:
            v.record_stream(cur)
        total = photo_root.detach() + reg_root.detach()
        if l2_grads is not None:
            total = total + loss_dict["l2"].reshape(())
        loss_dict["loss"] = total
        return loss_dict["loss"]
