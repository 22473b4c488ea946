"""Stand-alone ``lbs_extra``: drop-in for ``sings/rec/utils/body_model/lbs.py:16-74`` with the reference's
signature and return values, for callers that keep ``SinGS.forward`` (sings_hybrid.py:400-406, :526-533) as it is:

    from sings_amd.lbs import lbs_extra            # instead of  from ..utils.body_model.lbs import lbs_extra

``T = (W @ A.view(B, J, 16)).view(B, N, 4, 4)`` and ``verts = (T @ [v;1])[:, :, :3, 0]`` are ONE HIP kernel per batch
element (``sg_lbs_forward``: the W.A contraction on the matrix cores, exact fp32), the transpose for autograd another
(``sg_lbs_backward``: dL/dA = W^T . dT on the matrix cores, fixed-order reduction, no atomics; dL/dv).  The fused path
(``sings_amd.renderer.get_render_pkg_fused``) never materialises T; this op is for unmodified call sites.

``pose`` only feeds the pose-corrective blend shapes, which SinGS disables (``disable_posedirs=True``: ctor default
sings_hybrid.py:57); with ``disable_posedirs=False`` the offsets are computed as the reference does (Rodrigues + one
matmul, in torch on the GPU) before the kernel.  ``lbs_weights`` must not require a gradient (the reference detaches
them, sings_hybrid.py:724).
"""
import ctypes as C

import torch

from . import _lib
from .body import rodrigues


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Lbs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, v, lbs_weights):
        if not v.is_cuda:
            raise RuntimeError("sings_amd.lbs: tensors must live on the GPU (no CPU fallback)")
        lib = _lib.load()
        A = A.contiguous().float(); v = v.contiguous().float(); w = lbs_weights.contiguous().float()
        N, J = int(v.shape[0]), int(A.shape[0])
        if w.shape != (N, J):
            raise RuntimeError(f"lbs_weights must be [{N},{J}], got {tuple(w.shape)}")
        T = torch.empty((N, 4, 4), dtype=torch.float32, device=v.device)
        verts = torch.empty((N, 3), dtype=torch.float32, device=v.device)
        with torch.cuda.device(v.device):
            st = C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
            _lib.check(lib.sg_lbs_forward(N, J, _ptr(w), _ptr(A), _ptr(v), _ptr(T), _ptr(verts), st), "lbs forward")
        ctx.save_for_backward(A, v, w)
        return verts, T

    @staticmethod
    def backward(ctx, dverts, dT):
        lib = _lib.load()
        A, v, w = ctx.saved_tensors
        N, J = int(v.shape[0]), int(A.shape[0])
        dev = v.device
        dverts = None if dverts is None else dverts.contiguous().float()
        dT = None if dT is None else dT.contiguous().float()
        dv = torch.empty_like(v)
        dA = torch.empty((J, 4, 4), dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.sg_skin_ws_floats(N)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_lbs_backward(N, J, _ptr(w), _ptr(A), _ptr(v), _ptr(dT), _ptr(dverts), _ptr(ws), _ptr(dv), _ptr(dA), st),
                       "lbs backward")
        return dA, dv, None


def lbs_extra(A, v_shaped, posedirs, lbs_weights, pose, disable_posedirs=False, pose2rot=True):
    """-> (verts [B,N,3], A, T [B,N,4,4], v_posed, v_shaped), as lbs.py:16-74."""
    if lbs_weights.requires_grad:
        raise RuntimeError("sings_amd.lbs.lbs_extra: lbs_weights with requires_grad are not supported (the reference detaches them)")
    batch_size = A.shape[0]
    if disable_posedirs:
        v_posed = v_shaped                                      # (the reference adds zeros_like(v_shaped))
    else:
        ident = torch.eye(3, dtype=A.dtype, device=A.device)
        if pose2rot:
            rot_mats = rodrigues(pose.view(-1, 3)).view(batch_size, -1, 3, 3)
        else:
            rot_mats = pose.view(batch_size, -1, 3, 3)
        pose_feature = (rot_mats[:, 1:, :, :] - ident).view(batch_size, -1)
        v_posed = torch.matmul(pose_feature, posedirs).view(batch_size, -1, 3) + v_shaped
    vb = v_posed.expand(batch_size, -1, -1)
    outs = [_Lbs.apply(A[b].reshape(-1, 4, 4), vb[b], lbs_weights) for b in range(batch_size)]
    verts = torch.stack([o[0] for o in outs]); T = torch.stack([o[1] for o in outs])
    return verts, A, T, v_posed, v_shaped
