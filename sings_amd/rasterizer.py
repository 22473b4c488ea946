"""Drop-in replacement of the ``diff_gaussian_rasterization`` Python package on MI355X.

Same public surface the reference imports (sings/rec/renderer/gs_renderer_single.py:6-9,
gs_renderer_multiple.py:6-9; package = install_all.sh:22): ``GaussianRasterizationSettings``
(12-field NamedTuple), ``GaussianRasterizer`` (nn.Module, returns ``(color, radii)``),
``rasterize_gaussians`` and the ``_RasterizeGaussians`` autograd.Function, with the upstream
argument order and gradient order ``(means3D, means2D, sh, colors_precomp, opacities, scales,
rotations, cov3Ds_precomp, None)``.  The compute is the hand-written gfx950 library behind the
C ABI of include/sings_hip.h -- there is no CPU / eager fallback.
"""
import ctypes as C
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


# capacity (pairs) remembered per device so that steady-state calls never re-run
_capacity_hint = {}

# Upstream's forward blocks the host once per call (D2H read of the pair count in the middle of it); here that read sits
# at the END of the forward and the call is re-run with a larger workspace when the count exceeds the capacity.  In
# DEFERRED mode the forward does not read the count at all: the host never waits for the GPU inside a step (the Python
# side can run ahead, and a whole step can be captured into a HIP graph).  The price: a frame whose count exceeds the
# capacity renders the background and gets zero gradients (the kernels never follow partly written lists) until the
# caller polls ``check_deferred_overflow()`` -- e.g. every few hundred steps, or after densification -- which raises
# and grows the capacity for the following calls.
_deferred = {"on": False}
_pending = {}                          # device index -> (binning workspace, capacity) of the last deferred forward


def set_deferred_overflow_check(on=True, capacity_pairs=None, device=None):
    """Switch the forward of both autograd functions between the synchronous pair-count check (default) and the
    deferred one.  ``capacity_pairs`` presets the capacity (otherwise the last synchronous call's hint is used)."""
    _deferred["on"] = bool(on)
    if capacity_pairs is not None:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _capacity_hint[dev.index] = int(capacity_pairs)


def check_deferred_overflow(device=None):
    """Pair count of the last deferred forward on ``device`` (one D2H read = one host synchronisation).  Raises
    RuntimeError if it exceeded the capacity, after growing the capacity used by the following calls."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index not in _pending:
        return None
    binning, cap = _pending[dev.index]
    nr = C.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().sg_read_num_rendered(_ptr(binning), C.byref(nr), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "read R")
    R = int(nr.value)
    if R > cap:
        _capacity_hint[dev.index] = int(R * 1.25) + 1024
        raise RuntimeError(f"sings_amd: the last deferred forward produced {R} (tile, Gaussian) pairs for a capacity of {cap}: "
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
    s.prefiltered = int(bool(rs.prefiltered)); s.debug = int(bool(rs.debug)); s.reserved = 0
    s.bg = bg.data_ptr(); s.viewmatrix = vm.data_ptr(); s.projmatrix = pm.data_ptr(); s.campos = cp.data_ptr()
    return s


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
        T = ((W + 15) // 16) * ((H + 15) // 16)
        cap = max(_capacity_hint.get(dev.index, 0), 4 * P + T, 1 << 16)
        deferred = _deferred["on"]
        with torch.cuda.device(dev):
            while True:
                L = _lib.layout(P, W, H, cap)
                geom = torch.empty(L.geom_bytes, dtype=torch.uint8, device=dev)
                binning = torch.empty(L.bin_bytes, dtype=torch.uint8, device=dev)
                img = torch.empty(L.img_bytes, dtype=torch.uint8, device=dev)
                nr = C.c_int64(0)
                _lib.check(lib.sg_rasterize_forward(
                    C.byref(s), P, _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
                    _ptr(rotations), _ptr(cov3Ds_precomp), _ptr(geom), _ptr(binning), cap, _ptr(img),
                    _ptr(color), _ptr(radii), int(bool(write_point_keys)), None if deferred else C.byref(nr), stream), "forward")
                R = None if deferred else int(nr.value)
                if deferred or R <= cap:
                    break
                cap = int(R * 1.25) + 1024          # workspace too small: grow and re-run
        if deferred:
            _pending[dev.index] = (binning, cap)
        else:
            _capacity_hint[dev.index] = max(int(R * 1.25) + 1024, _capacity_hint.get(dev.index, 0) * 3 // 4)
        ctx.raster_settings = rs
        ctx.num_rendered = R
        ctx.capacity = cap
        ctx.sh_coeffs = M
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
        keep = []
        s = _settings_struct(rs, dev, ctx.sh_coeffs, keep)
        g = _f32(grad_out_color, "grad_out_color", dev)

        def e(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        dmeans3D, dmeans2D, dopac = e(P, 3), e(P, 3), e(P, 1)
        dsh = e(P, ctx.sh_coeffs, 3) if has_sh else None
        dcol = e(P, 3) if has_col else None
        dscales = e(P, 3) if has_scale else None
        drots = e(P, 4) if has_scale else None
        dcov = e(P, 6) if has_cov else None
        with torch.cuda.device(dev):
            L = _lib.layout(P, W, H, ctx.capacity)
            bwd_ws = torch.empty(L.bwd_bytes, dtype=torch.uint8, device=dev)
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
