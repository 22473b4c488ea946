"""L1 + SSIM photometric loss of one rendered view through the C ABI (sg_photo_loss).

Mirrors the L1 / SSIM part of ``HumanSceneLoss.forward`` (sings/rec/losses/loss.py:46-69) together with the
renderer's final clamp (gs_renderer_single.py:96):

    pred = clamp(raw, 0, 1);  gt = rgb * mask + bg * (1 - mask)
    loss_dict['l1']   = l1_w   * |pred - gt|.sum() / mask.sum()
    loss_dict['ssim'] = ssim_w * (1 - ssim(pred, gt)) * mask.sum() / (H * W)

``photometric_loss`` takes the UNCLAMPED rasterizer output (the clamp is fused) and returns the reference's
``loss_dict`` / ``extras_dict`` entries; both loss tensors carry autograd back to ``raw``.  Forward and gradient are ONE kernel
(csrc/sg_loss.hip): a forward whose input requires grad computes the gradient for unit upstream weights in the same march and the
backward multiplies it by the upstream scalar when both terms share it (the usual ``l1 + ssim`` under one sum); with different
weights for the two terms the gradient kernel runs again with them as device scalars (sg_photo_loss_backward).  No host
synchronisation anywhere.  LPIPS (a VGG network) is not provided.
No CPU fallback: the HIP library must be present.
"""
import ctypes as C

import torch

from . import _lib
from .rasterizer import _ptr


def _same_upstream(a, b):
    """Both loss terms received the SAME upstream gradient -- the same memory, not merely equal values (nothing is read back): the two
    outputs sit under one sum (autograd hands each consumer its own tensor OBJECT for the one buffer ``_AddScalars`` / ``_SumFrames``
    returned twice)."""
    return (a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
            and a.dtype == b.dtype)


_UNIT = {}                             # data_ptr -> weak reference to a 0-dim tensor its owner vouches holds 1.0 and never writes


def register_unit_scalar(t):
    """``t``: a 0-dim float tensor that holds 1.0 for as long as it lives (the seed a caller hands ``torch.autograd.grad`` for its
    loss root: sings_amd.train_step).  A backward pass whose two upstream gradients ARE that tensor returns the gradient the forward
    computed as it is -- no ``unit * 1.0`` launch (a broadcast multiplication of a 1080p image: 12 us on the step's main chain)."""
    import weakref
    _UNIT[t.data_ptr()] = weakref.ref(t)
    return t


def _is_unit(g):
    r = _UNIT.get(g.data_ptr())
    t = r() if r is not None else None
    if t is None:
        _UNIT.pop(g.data_ptr(), None)
        return False
    return t.data_ptr() == g.data_ptr() and g.dtype == t.dtype and g.storage_offset() == t.storage_offset()


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, gt_rgb, mask, bg, l1_w, ssim_w, want_images):
        if not raw.is_cuda:
            raise RuntimeError("sings_amd.photo_loss: tensors must live on the GPU (no CPU fallback)")
        lib = _lib.load()
        raw = raw.contiguous().float(); gt_rgb = gt_rgb.contiguous().float()
        mask = mask.contiguous().float(); bg = bg.contiguous().float()
        if raw.dim() != 3 or raw.shape[0] != 3 or gt_rgb.shape != raw.shape or mask.numel() != raw.shape[1] * raw.shape[2]:
            raise ValueError("photometric_loss: raw / gt_rgb must be [3,H,W] and mask [H,W] or [1,H,W]")
        H, W = int(raw.shape[1]), int(raw.shape[2])
        dev = raw.device
        ws = torch.empty(int(lib.sg_photo_loss_ws_bytes(W, H)), dtype=torch.uint8, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        pred = torch.empty_like(raw) if want_images else None
        gt = torch.empty_like(raw) if want_images else None
        # Round 6: forward and gradient are ONE kernel (sg_loss.hip), so a forward that will be differentiated computes the gradient
        # for unit upstream weights right away; the usual backward -- both terms under one sum, i.e. the SAME upstream tensor for
        # both -- is then one multiplication by that scalar instead of a second march over the image.
        unit = torch.empty_like(raw) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_photo_loss(W, H, float(l1_w), float(ssim_w), _ptr(raw), _ptr(gt_rgb), _ptr(mask), _ptr(bg),
                                         _ptr(ws), _ptr(pred), _ptr(gt), _ptr(losses), None, _ptr(unit), stream),
                       "photo loss")
        # the mask partial sums stay in `ws`; a backward with DIFFERENT upstream weights for the two terms runs the gradient kernel
        # again with them as device scalars -- nothing is read back to compare them (a host synchronisation per step)
        ctx.unit_grad = unit
        ctx.save_for_backward(raw, gt_rgb, mask, bg, ws)
        ctx.args = (float(l1_w), float(ssim_w), W, H)
        ctx.set_materialize_grads(False)       # the two report-only outputs get None, not a zero tensor each (two fill launches)
        outs = (losses[0], losses[1], losses[2], losses[3])
        if want_images:
            ctx.mark_non_differentiable(pred, gt)
            return outs + (pred, gt)
        return outs

    @staticmethod
    def backward(ctx, g_l1, g_ssim, g_raw_l1=None, g_ssim_mean=None, *unused):
        raw, gt_rgb, mask, bg, ws = ctx.saved_tensors
        l1_w, ssim_w, W, H = ctx.args
        lib = _lib.load()
        dev = raw.device
        unit, ctx.unit_grad = ctx.unit_grad, None
        if unit is not None and _same_upstream(g_l1, g_ssim):
            return (unit if _is_unit(g_l1) else unit * g_l1.reshape(()).float()), None, None, None, None, None, None
        if g_l1 is None or g_ssim is None:
            zero = torch.zeros((), dtype=torch.float32, device=dev)
            g_l1 = zero if g_l1 is None else g_l1
            g_ssim = zero if g_ssim is None else g_ssim
        up = torch.stack([g_l1.reshape(()).float(), g_ssim.reshape(()).float()])
        out = torch.empty_like(raw)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_photo_loss_backward(W, H, l1_w, ssim_w, _ptr(raw), _ptr(gt_rgb), _ptr(mask), _ptr(bg), _ptr(ws),
                                                  _ptr(up), _ptr(out), stream), "photo loss backward")
        return out, None, None, None, None, None, None


def photometric_loss(raw_image, gt_rgb, mask, bg_color, l1_w=0.8, ssim_w=0.2, return_images=False):
    """-> (loss_dict, extras_dict) with the reference's keys 'l1', 'ssim' (weighted, differentiable w.r.t. raw_image)
    and, when ``return_images``, 'pred_img' / 'gt_img'.  Default weights: human.loss.l1_w / ssim_w (defaults/config.py:89-90).
    The unweighted L1 and the mean SSIM are returned under 'l1_raw' / 'ssim_mean' in extras_dict (detached)."""
    out = _PhotoLoss.apply(raw_image, gt_rgb, mask, bg_color, l1_w, ssim_w, bool(return_images))
    loss_dict = {}
    if l1_w > 0.0:
        loss_dict["l1"] = out[0]
    if ssim_w > 0.0:
        loss_dict["ssim"] = out[1]
    extras = {"l1_raw": out[2].detach(), "ssim_mean": out[3].detach()}
    if return_images:
        extras["pred_img"], extras["gt_img"] = out[4], out[5]
    return loss_dict, extras


class _PhotoLossFrames(torch.autograd.Function):
    """K frames at once: ``raw`` [K,3,H,W]; ``gt_rgb`` [K,3,H,W] or [3,H,W]; ``mask`` [K,H,W] / [K,1,H,W] or [H,W].  Returns the four
    loss scalars per frame as [K] tensors; backward takes a pair of upstream weights PER FRAME (sg_photo_loss_backward_frames)."""

    @staticmethod
    def forward(ctx, raw, gt_rgb, mask, bg, l1_w, ssim_w):
        if not raw.is_cuda:
            raise RuntimeError("sings_amd.photo_loss: tensors must live on the GPU (no CPU fallback)")
        lib = _lib.load()
        raw = raw.contiguous().float(); gt_rgb = gt_rgb.contiguous().float()
        mask = mask.contiguous().float(); bg = bg.contiguous().float()
        if raw.dim() != 4 or raw.shape[1] != 3 or not 1 <= raw.shape[0] <= _lib.MAX_FRAMES:
            raise ValueError(f"photometric_loss_frames: raw must be [K,3,H,W] with K in 1..{_lib.MAX_FRAMES}")
        K, H, W = int(raw.shape[0]), int(raw.shape[2]), int(raw.shape[3])
        hw = H * W
        if gt_rgb.numel() not in (3 * hw, K * 3 * hw) or mask.numel() not in (hw, K * hw):
            raise ValueError("photometric_loss_frames: gt_rgb must be [K,3,H,W] or [3,H,W], mask [K,H,W] or [H,W]")
        gt_stride = 3 * hw if gt_rgb.numel() == K * 3 * hw and K > 1 else 0
        mask_stride = hw if mask.numel() == K * hw and K > 1 else 0
        dev = raw.device
        ws = torch.empty(K * int(lib.sg_photo_loss_ws_bytes(W, H)), dtype=torch.uint8, device=dev)
        losses = torch.empty((K, 4), dtype=torch.float32, device=dev)
        unit = torch.empty_like(raw) if ctx.needs_input_grad[0] else None       # (gradient for unit upstream weights: see _PhotoLoss)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_photo_loss_frames(K, W, H, float(l1_w), float(ssim_w), _ptr(raw), _ptr(gt_rgb), gt_stride, _ptr(mask),
                                                mask_stride, _ptr(bg), _ptr(ws), None, None, _ptr(losses), None, _ptr(unit), stream),
                       "photo loss (frames)")
        ctx.unit_grad = unit
        ctx.save_for_backward(raw, gt_rgb, mask, bg, ws)
        ctx.args = (float(l1_w), float(ssim_w), K, W, H, gt_stride, mask_stride)
        ctx.set_materialize_grads(False)
        return losses[:, 0], losses[:, 1], losses[:, 2], losses[:, 3]

    @staticmethod
    def backward(ctx, g_l1, g_ssim, g_raw_l1=None, g_ssim_mean=None):
        raw, gt_rgb, mask, bg, ws = ctx.saved_tensors
        l1_w, ssim_w, K, W, H, gt_stride, mask_stride = ctx.args
        lib = _lib.load()
        dev = raw.device
        unit, ctx.unit_grad = ctx.unit_grad, None
        if unit is not None and _same_upstream(g_l1, g_ssim):
            return (unit if _is_unit(g_l1) else unit * g_l1.reshape(K, 1, 1, 1).float()), None, None, None, None, None
        zero = torch.zeros(K, dtype=torch.float32, device=dev)
        up = torch.stack([zero if g_l1 is None else g_l1.reshape(K).float(), zero if g_ssim is None else g_ssim.reshape(K).float()], 1)
        up = up.contiguous()                                                   # [K,2]: a pair of weights per frame, device memory
        out = torch.empty_like(raw)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sg_photo_loss_backward_frames(K, W, H, l1_w, ssim_w, _ptr(raw), _ptr(gt_rgb), gt_stride, _ptr(mask), mask_stride,
                                                         _ptr(bg), _ptr(ws), _ptr(up), 2, _ptr(out), stream), "photo loss backward (frames)")
        return out, None, None, None, None, None


def photometric_loss_frames(raw_images, gt_rgb, mask, bg_color, l1_w=0.8, ssim_w=0.2):
    """The K-frame form of ``photometric_loss`` (three launches forward, one backward for the K frames): ``raw_images`` [K,3,H,W]
    (e.g. ``get_render_pkgs_fused(...)['render_raw']``), ``gt_rgb`` [K,3,H,W] or one [3,H,W] target, ``mask`` likewise.
    -> (loss_dict, extras_dict): 'l1' / 'ssim' are [K] tensors (weighted, differentiable; ``sum()`` them for the step's loss),
    'l1_raw' / 'ssim_mean' [K] detached."""
    out = _PhotoLossFrames.apply(raw_images, gt_rgb, mask, bg_color, l1_w, ssim_w)
    loss_dict = {}
    if l1_w > 0.0:
        loss_dict["l1"] = out[0]
    if ssim_w > 0.0:
        loss_dict["ssim"] = out[1]
    return loss_dict, {"l1_raw": out[2].detach(), "ssim_mean": out[3].detach()}


class PhotoLossEngine:
    """Pre-allocated variant for training loops / bench.py: no allocation, no synchronisation per call.  ``K`` > 1: the losses
    and gradients of K frames in three launches (``sg_photo_loss_frames``): ``raw`` [K,3,H,W]; ``gt_rgb`` [K,3,H,W] or one
    [3,H,W] target for all frames, ``mask`` likewise; ``losses`` [K,4], ``grad`` [K,3,H,W]."""

    def __init__(self, W, H, device, l1_w=0.8, ssim_w=0.2, K=1):
        self.lib = _lib.load()
        self.W, self.H, self.l1_w, self.ssim_w, self.K = int(W), int(H), float(l1_w), float(ssim_w), int(K)
        self.dev = torch.device(device)
        self.ws = torch.empty(self.K * int(self.lib.sg_photo_loss_ws_bytes(self.W, self.H)), dtype=torch.uint8, device=self.dev)
        shape = (lambda *s: s) if self.K == 1 else (lambda *s: (self.K,) + s)
        self.losses = torch.zeros(shape(4), dtype=torch.float32, device=self.dev)
        self.grad = torch.empty(shape(3, self.H, self.W), dtype=torch.float32, device=self.dev)

    def __call__(self, raw, gt_rgb, mask, bg):
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        hw = self.H * self.W
        gt_stride = 3 * hw if gt_rgb.numel() == self.K * 3 * hw and self.K > 1 else 0
        mask_stride = hw if mask.numel() == self.K * hw and self.K > 1 else 0
        _lib.check(self.lib.sg_photo_loss_frames(self.K, self.W, self.H, self.l1_w, self.ssim_w, _ptr(raw), _ptr(gt_rgb), gt_stride,
                                                 _ptr(mask), mask_stride, _ptr(bg), _ptr(self.ws), None, None, _ptr(self.losses),
                                                 None, _ptr(self.grad), stream), "photo loss")
        return self.grad
