"""Fused "posed-frame" operator: SMPL-LBS deformation of canonical Gaussians + rasterization.

Opt-in replacement of the LBS block of ``SinGS.forward`` (sings/rec/models/sings_hybrid.py:398-428)
followed by ``get_render_pkg`` (sings/rec/renderer/gs_renderer_single.py:12-42): the caller hands over
canonical means / rotations / scales, the skinning weights and the cano->pose joint transforms
``A_cano2pose = A_t2pose @ inv_A_t2cano`` (:398-399); posed means, rotation quaternions and the
blended T[N,4,4] never reach HBM.  Gradients flow to xyz_canon, rotmat_canon, scales, opacity, shs,
``A`` (hence body_pose / global_orient through the SMPL chain in torch) and ``transl``.
"""
import ctypes as C

import torch

from . import _lib
from . import rasterizer as _rz
from .rasterizer import _f32, _ptr, _settings_struct


def _skin_struct(dev, xyz_canon, rotmat_canon, lbs_weights, A, smpl_scale, transl, ext_tfs, keep):
    """rotmat_canon: None (isotropic), [N,9] / [N,3,3] rotation matrices, or [N,6] in the decoder's 6-D form
    (rotation_6d_to_matrix then runs inside the kernels)."""
    k = _lib.SgSkinInputs()
    A = _f32(A, "A", dev).reshape(-1, 16)
    k.J = int(A.shape[0])
    k.rot_format = 1 if (rotmat_canon is not None and rotmat_canon.shape[-1] == 6 and rotmat_canon.dim() == 2) else 0
    lbs_weights = _f32(lbs_weights, "lbs_weights", dev)
    if lbs_weights.shape != (xyz_canon.shape[0], k.J):
        raise RuntimeError(f"lbs_weights must be [N,{k.J}], got {tuple(lbs_weights.shape)}")
    ts = [xyz_canon, rotmat_canon, lbs_weights, A,
          None if smpl_scale is None else _f32(smpl_scale, "smpl_scale", dev).reshape(-1),
          None if transl is None else _f32(transl, "transl", dev).reshape(-1)]
    if ext_tfs is not None:
        tr, rm, sc = ext_tfs
        ts += [_f32(tr, "ext trans", dev).reshape(-1), _f32(rm, "ext rotmat", dev).reshape(-1),
               _f32(sc, "ext scale", dev).reshape(-1)]
    else:
        ts += [None, None, None]
    keep.extend(ts)
    for name, t in zip(("xyz_canon", "rot_canon", "lbs_weights", "A", "smpl_scale", "transl", "ext_trans", "ext_rot",
                        "ext_scale"), ts):
        setattr(k, name, None if t is None else t.data_ptr())
    return k


class _RasterizeSkinnedGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_canon, rotmat_canon, scales, opacities, shs, A, transl, lbs_weights, smpl_scale, ext_tfs,
                raster_settings, return_posed, means2D=None):
        lib = _lib.load()
        dev = xyz_canon.device
        if dev.type != "cuda":
            raise RuntimeError("sings_amd fused LBS+raster runs on the MI355X only; there is no CPU fallback")
        rs = raster_settings
        xyz_canon = _f32(xyz_canon, "xyz_canon", dev)
        rot_shape = None if rotmat_canon is None else tuple(rotmat_canon.shape)
        if rotmat_canon is not None:
            rotmat_canon = _f32(rotmat_canon, "rotmat_canon", dev)
            if rotmat_canon.shape[-2:] == (3, 3):
                rotmat_canon = rotmat_canon.reshape(-1, 9)
            elif rotmat_canon.dim() != 2 or rotmat_canon.shape[1] not in (6, 9):
                raise RuntimeError(f"rotmat_canon must be [N,3,3], [N,9] or [N,6] (6-D form), got {tuple(rotmat_canon.shape)}")
        scales = _f32(scales, "scales", dev); opacities = _f32(opacities, "opacities", dev); shs = _f32(shs, "shs", dev)
        P = int(xyz_canon.shape[0]); H, W = int(rs.image_height), int(rs.image_width); M = int(shs.shape[1])
        keep = []
        s = _settings_struct(rs, dev, M, keep)
        k = _skin_struct(dev, xyz_canon, rotmat_canon, lbs_weights, A, smpl_scale, transl, ext_tfs, keep)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        pxyz, pq, psc = (e(P, 3), e(P, 4), e(P, 3)) if return_posed else (None, None, None)
        cap, sync, sig = _rz._forward_plan(dev, P, W, H)     # see rasterizer.set_overflow_check
        hint_key = (dev.index,) + sig
        rows = not rs.debug and _rz._rows_ok.get(hint_key, False)    # learned by a "sync" forward of this shape (rasterizer.py)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if sync and not rs.debug:
                _rz._arm_early_count(s, dev)
            while True:
                s.flags = _lib.FLAG_LONG_ROWS if rows else 0
                L = _lib.layout(P, W, H, cap)
                geom = torch.empty(L.geom_bytes, dtype=torch.uint8, device=dev)
                binning = torch.empty(L.bin_bytes, dtype=torch.uint8, device=dev)
                img = torch.empty(L.img_bytes, dtype=torch.uint8, device=dev)
                nr = C.c_int64(0)
                _lib.check(lib.sg_skinned_forward(C.byref(s), P, C.byref(k), _ptr(shs), _ptr(opacities), _ptr(scales),
                                                  _ptr(geom), _ptr(binning), cap, _ptr(img), _ptr(color), _ptr(radii),
                                                  _ptr(pxyz), _ptr(pq), _ptr(psc), C.byref(nr) if sync else None, stream),
                           "skinned forward")
                R = int(nr.value) if sync else None
                if sync and rows and R == _lib.NUM_RENDERED_LONG_LIST:       # a list outgrew its row: this frame again, without the hint
                    _rz._rows_ok[hint_key] = rows = False
                    continue
                if not sync or R <= cap:
                    break
                cap = int(R * _rz._HEADROOM) + 1024
            if sync:
                _rz._forward_done_sync(dev, R, sig)
                _rz._learn_hints(s, hint_key)
            else:
                _rz._after_forward(dev, binning, cap)
        ctx.rs, ctx.cap, ctx.M, ctx.num_rendered = rs, cap, M, R
        ctx.has_rot = rotmat_canon is not None
        ctx.has_ext = ext_tfs is not None
        ctx.rot_shape = rot_shape
        ctx.has_m2d = means2D is not None
        ctx.return_posed = return_posed
        ctx.aux = (lbs_weights, smpl_scale, A.shape, None if transl is None else transl.shape)
        z = torch.empty(0, device=dev)
        ctx.save_for_backward(xyz_canon, rotmat_canon if ctx.has_rot else z, scales, opacities, shs,
                              _f32(A, "A", dev), transl if transl is not None else z, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        if return_posed:
            ctx.mark_non_differentiable(psc)
            return color, radii, pxyz, pq, psc
        return color, radii

    @staticmethod
    def backward(ctx, g_color, _g_radii=None, g_pxyz=None, g_pq=None, _g_psc=None):
        lib = _lib.load()
        if ctx.has_ext:
            raise RuntimeError("ext_tfs are forward-only (the reference applies them under no_grad: anim_avatar.py:27)")
        xyz_canon, rotmat_canon, scales, opacities, shs, A, transl, radii, geom, binning, img = ctx.saved_tensors
        lbs_weights, smpl_scale, A_shape, transl_shape = ctx.aux
        dev = xyz_canon.device
        rs = ctx.rs
        P = int(xyz_canon.shape[0]); H, W = int(rs.image_height), int(rs.image_width)
        keep = []
        s = _settings_struct(rs, dev, ctx.M, keep)
        k = _skin_struct(dev, xyz_canon, rotmat_canon if ctx.has_rot else None, lbs_weights, A, smpl_scale,
                         transl if transl_shape is not None else None, None, keep)
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        d_xyz, d_scales, d_op, d_sh, d_m2d = e(P, 3), e(P, 3), e(P, 1), e(P, ctx.M, 3), e(P, 3)
        d_rot = e(P, int(rotmat_canon.shape[1])) if ctx.has_rot else None
        d_A = e(k.J, 16); d_tr = e(3)
        g_color = _f32(g_color, "grad_out_color", dev)
        g_pxyz = None if g_pxyz is None else _f32(g_pxyz, "grad posed xyz", dev)
        g_pq = None if g_pq is None else _f32(g_pq, "grad posed rotq", dev)
        with torch.cuda.device(dev):
            L = _lib.layout(P, W, H, ctx.cap)
            bwd_ws = torch.empty(L.bwd_bytes, dtype=torch.uint8, device=dev)
            skin_ws = e(int(lib.sg_skin_ws_floats(P)))
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_skinned_backward(
                C.byref(s), P, C.byref(k), _ptr(shs), _ptr(opacities), _ptr(scales), _ptr(radii), _ptr(geom),
                _ptr(binning), ctx.cap, _ptr(img), _ptr(bwd_ws), _ptr(skin_ws), _ptr(g_color), _ptr(g_pxyz), _ptr(g_pq),
                _ptr(d_xyz), _ptr(d_rot), _ptr(d_scales), _ptr(d_op), _ptr(d_sh), _ptr(d_m2d), _ptr(d_A), _ptr(d_tr),
                stream), "skinned backward")
        return (d_xyz, None if d_rot is None else d_rot.view(ctx.rot_shape), d_scales, d_op.view_as(opacities), d_sh,
                d_A.view(A_shape), None if transl_shape is None else d_tr.view(transl_shape), None, None, None, None, None,
                d_m2d if ctx.has_m2d else None)


def rasterize_skinned_gaussians(xyz_canon, rotmat_canon, scales, opacities, shs, lbs_weights, A_cano2pose,
                                raster_settings, smpl_scale=None, transl=None, ext_tfs=None, return_posed=False,
                                means2D=None):
    """color, radii[, posed_xyz, posed_rotq, posed_scales] = fused LBS + rasterization of one posed frame.

    ``rotmat_canon=None`` means isotropic Gaussians (identity canonical rotation, sings_hybrid.py:358-361); a ``[N,6]``
    tensor is the geometry decoder's 6-D output: ``rotation_6d_to_matrix`` (sings_hybrid.py:356-357) is then evaluated
    inside the kernels, forward and backward, and the gradient comes back in 6-D form.
    ``means2D``: optional ``[N,3]`` tensor with ``requires_grad`` (values unused) that receives the screen-space
    gradient the densifier reads as ITS gradient -- the reference's ``viewspace_points`` idiom
    (gs_renderer_single.py:50-56, consumer sings_hybrid.py:1013-1015), one holder per call, so several views of a
    step keep their own statistics.
    """
    return _RasterizeSkinnedGaussians.apply(xyz_canon, rotmat_canon, scales, opacities, shs, A_cano2pose, transl,
                                            lbs_weights, smpl_scale, ext_tfs, raster_settings, return_posed, means2D)


class _RasterizeSkinnedFrames(torch.autograd.Function):
    """K posed frames of the SAME canonical Gaussians in ONE call per direction (the ``*_frames`` entry points of
    include/sings_hip.h): what ``SinGS.forward_chunk`` (sings_hybrid.py:474-569) + the render loop of gs_trainer.py:684-714 do for
    a chunk, as one differentiable operator.  The gradients of the canonical Gaussians are the SUM over the K frames (formed inside
    the per-Gaussian backward kernel, in frame order); dL/dA and dL/dtransl are per frame.  The pair counts of the K frames are read
    before the call returns (one strided copy + stream synchronisation): a frame that would overflow is never returned, the
    workspaces grow and the call is repeated.  With ``set_overflow_check("deferred")`` or inside a HIP-graph capture nothing is
    read on the host: the largest (pair count, overflow flag) of the K frames is folded into the device-side accumulator that
    ``check_deferred_overflow()`` polls, exactly as the single-frame operator does."""

    @staticmethod
    def forward(ctx, xyz_canon, rotmat_canon, scales, opacities, shs, A, transl, lbs_weights, smpl_scale, raster_settings,
                means2D=None):
        from .engine import _frame_batch
        lib = _lib.load()
        dev = xyz_canon.device
        if dev.type != "cuda":
            raise RuntimeError("sings_amd fused LBS+raster runs on the MI355X only; there is no CPU fallback")
        rs = raster_settings
        xyz_canon = _f32(xyz_canon, "xyz_canon", dev)
        rot_shape = None if rotmat_canon is None else tuple(rotmat_canon.shape)
        if rotmat_canon is not None:
            rotmat_canon = _f32(rotmat_canon, "rotmat_canon", dev)
            if rotmat_canon.shape[-2:] == (3, 3):
                rotmat_canon = rotmat_canon.reshape(-1, 9)
            elif rotmat_canon.dim() != 2 or rotmat_canon.shape[1] not in (6, 9):
                raise RuntimeError(f"rotmat_canon must be [N,3,3], [N,9] or [N,6] (6-D form), got {tuple(rotmat_canon.shape)}")
        scales = _f32(scales, "scales", dev); opacities = _f32(opacities, "opacities", dev); shs = _f32(shs, "shs", dev)
        A = _f32(A, "A", dev)
        K = int(A.shape[0])
        if A.dim() < 3 or not 1 <= K <= _lib.MAX_FRAMES:
            raise RuntimeError(f"A must be [K,J,4,4] / [K,J,16] with K in 1..{_lib.MAX_FRAMES}, got {tuple(A.shape)}")
        A16 = A.reshape(K, -1, 16)
        tstride = 0
        if transl is not None:
            transl = _f32(transl, "transl", dev)
            if transl.numel() == 3 * K and K > 1:
                tstride = 3
            elif transl.numel() != 3:
                raise RuntimeError(f"transl must be [{K},3] or [3], got {tuple(transl.shape)}")
        vm = rs.viewmatrix
        cam_stride = 1 if vm.dim() == 3 else 0
        if cam_stride and (vm.shape[0] != K or rs.projmatrix.shape[0] != K or rs.campos.shape[0] != K):
            raise RuntimeError(f"per-frame cameras must be stacked [{K},4,4] / [{K},3]")
        P = int(xyz_canon.shape[0]); H, W = int(rs.image_height), int(rs.image_width); M = int(shs.shape[1])
        keep = []
        s = _settings_struct(rs, dev, M, keep)
        s.flags = _lib.FLAG_THROUGHPUT
        k = _skin_struct(dev, xyz_canon, rotmat_canon, lbs_weights, A16[0], smpl_scale,
                         None if transl is None else transl.reshape(-1)[:3], None, keep)
        k.A = A16.data_ptr()
        k.transl = None if transl is None else transl.data_ptr()
        fb = _frame_batch(K, cam_stride, tstride)
        color = torch.empty((K, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((K, P), dtype=torch.int32, device=dev)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        # the same plan as the single-frame op (rasterizer.set_overflow_check): "sync" reads the K pair counts before returning,
        # "async" does so for the first call of a shape only and afterwards copies the K headers' (worst R, OR of the flags) to the
        # pinned ring without waiting, "deferred" / a graph capture folds them into the device-side accumulator
        cap, sync, sig = _rz._forward_plan(dev, P, W, H)
        hint_key = (dev.index,) + sig
        rows = not rs.debug and _rz._rows_ok.get(hint_key, False)    # learned by a "sync" single-frame forward of this shape
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            while True:
                s.flags = _lib.FLAG_THROUGHPUT | (_lib.FLAG_LONG_ROWS if rows else 0)
                L = _lib.SgLayout(); sizes = [C.c_size_t() for _ in range(4)]
                _lib.check(lib.sg_frames_layout(P, W, H, cap, K, C.byref(L), *[C.byref(x) for x in sizes]), "sg_frames_layout")
                geom = torch.empty(sizes[0].value, dtype=torch.uint8, device=dev)
                binning = torch.empty(sizes[1].value, dtype=torch.uint8, device=dev)
                img = torch.empty(sizes[2].value, dtype=torch.uint8, device=dev)
                nr = (C.c_int64 * K)()
                _lib.check(lib.sg_skinned_forward_frames(C.byref(s), C.byref(fb), P, C.byref(k), _ptr(shs), _ptr(opacities), _ptr(scales),
                                                         _ptr(geom), _ptr(binning), cap, _ptr(img), _ptr(color), _ptr(radii), None, None,
                                                         None, nr if sync else None, stream), "skinned forward (frames)")
                if not sync:
                    break
                if rows and min(int(v) for v in nr) == _lib.NUM_RENDERED_LONG_LIST:
                    _rz._rows_ok[hint_key] = rows = False
                    continue
                R = max(int(v) for v in nr)
                if R <= cap:
                    break
                cap = int(R * _rz._HEADROOM) + 1024
            if sync:
                _rz._forward_done_sync(dev, R, sig)
            else:
                # largest pair count and the OR of the K frames' flag words (a frame's "long list" bit must not hide another
                # frame's "overflow" bit: ADVICE r4) -> the asynchronous / deferred check of the single-frame op
                h = binning.view(K, L.bin_bytes)[:, :8].view(torch.int32)
                flags = (h[:, 1:2] & _rz._flag_bits(dev)).amax(0).sum().to(torch.int32)
                hdr = torch.stack([h[:, 0].amax(), flags]).contiguous()
                _rz._after_forward(dev, hdr.view(torch.uint8), cap)
                nr = [0] * K
        ctx.rs, ctx.cap, ctx.M, ctx.K, ctx.fb = rs, cap, M, K, (cam_stride, tstride)
        ctx.num_rendered = [int(v) for v in nr]
        ctx.has_rot = rotmat_canon is not None
        ctx.rot_shape = rot_shape
        ctx.has_m2d = means2D is not None
        ctx.aux = (lbs_weights, smpl_scale, A.shape, None if transl is None else transl.shape, sizes[3].value)
        z = torch.empty(0, device=dev)
        ctx.save_for_backward(xyz_canon, rotmat_canon if ctx.has_rot else z, scales, opacities, shs, A16,
                              transl if transl is not None else z, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, g_color, _g_radii=None):
        from .engine import _frame_batch
        lib = _lib.load()
        xyz_canon, rotmat_canon, scales, opacities, shs, A16, transl, radii, geom, binning, img = ctx.saved_tensors
        lbs_weights, smpl_scale, A_shape, transl_shape, bwd_bytes = ctx.aux
        dev = xyz_canon.device
        rs, K = ctx.rs, ctx.K
        P = int(xyz_canon.shape[0]); H, W = int(rs.image_height), int(rs.image_width)
        keep = []
        s = _settings_struct(rs, dev, ctx.M, keep)
        s.flags = _lib.FLAG_THROUGHPUT
        k = _skin_struct(dev, xyz_canon, rotmat_canon if ctx.has_rot else None, lbs_weights, A16[0], smpl_scale,
                         transl.reshape(-1)[:3] if transl_shape is not None else None, None, keep)
        k.A = A16.data_ptr()
        k.transl = transl.data_ptr() if transl_shape is not None else None
        fb = _frame_batch(K, *ctx.fb)
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        d_xyz, d_scales, d_op, d_sh, d_m2d = e(P, 3), e(P, 3), e(P, 1), e(P, ctx.M, 3), e(K, P, 3)
        d_rot = e(P, int(rotmat_canon.shape[1])) if ctx.has_rot else None
        d_A = e(K, k.J, 16); d_tr = e(K, 3)
        g_color = _f32(g_color, "grad_out_color", dev)
        with torch.cuda.device(dev):
            bwd_ws = torch.empty(bwd_bytes, dtype=torch.uint8, device=dev)
            skin_ws = e(int(lib.sg_skin_ws_floats_frames(P, K)))
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_rasterize_backward_records_frames(C.byref(s), C.byref(fb), P, _ptr(geom), _ptr(binning), ctx.cap, _ptr(img),
                                                                _ptr(bwd_ws), _ptr(g_color), stream), "skinned backward (records, frames)")
            _lib.check(lib.sg_skinned_backward_gaussians_frames(
                C.byref(s), C.byref(fb), P, C.byref(k), _ptr(shs), _ptr(opacities), _ptr(scales), _ptr(radii), _ptr(geom), _ptr(binning),
                ctx.cap, _ptr(bwd_ws), _ptr(skin_ws), 0, None, None, _ptr(d_xyz), _ptr(d_rot), _ptr(d_scales), _ptr(d_op), _ptr(d_sh),
                _ptr(d_m2d), _ptr(d_A), _ptr(d_tr), stream), "skinned backward (gaussians, frames)")
        if transl_shape is None:
            g_tr = None
        elif ctx.fb[1] == 3:
            g_tr = d_tr.view(transl_shape)
        else:
            g_tr = d_tr.sum(0).view(transl_shape)                 # one translation shared by the K frames
        return (d_xyz, None if d_rot is None else d_rot.view(ctx.rot_shape), d_scales, d_op.view_as(opacities), d_sh,
                d_A.view(A_shape), g_tr, None, None, None, d_m2d if ctx.has_m2d else None)


def rasterize_skinned_frames(xyz_canon, rotmat_canon, scales, opacities, shs, lbs_weights, A_cano2pose, raster_settings,
                             smpl_scale=None, transl=None, means2D=None):
    """color [K,3,H,W], radii [K,P] = fused LBS + rasterization of K posed frames of the same canonical Gaussians in one call.

    ``A_cano2pose`` [K,J,4,4]; ``transl`` [K,3] (per frame), [3] (shared) or None; ``raster_settings``: one camera, or
    ``viewmatrix`` / ``projmatrix`` [K,4,4] and ``campos`` [K,3] for a camera per frame.  ``means2D``: optional [K,N,3] holder
    (``requires_grad``) of the per-frame screen-space gradients.  Backward: canonical-Gaussian gradients summed over the K frames."""
    return _RasterizeSkinnedFrames.apply(xyz_canon, rotmat_canon, scales, opacities, shs, A_cano2pose, transl, lbs_weights,
                                         smpl_scale, raster_settings, means2D)
