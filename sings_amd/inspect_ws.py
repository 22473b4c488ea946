"""Typed views into the opaque workspaces of a forward call (tests / debugging only).

Runs the forward through the C ABI exactly as ``_RasterizeGaussians.forward`` does and returns
the intermediate state (projected records, sorted point list, upstream-format keys, tile ranges,
final_T, n_contrib) as tensors, so parity tests can compare integers bit for bit.
"""
import ctypes as C

import torch

from . import _lib
from .rasterizer import _settings_struct, _ptr


def forward_with_state(rs, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                       cov3D_precomp=None, capacity=None, flags=0):
    lib = _lib.load()
    dev = means3D.device
    P = int(means3D.shape[0]); H, W = int(rs.image_height), int(rs.image_width)
    M = int(shs.shape[1]) if shs is not None else 0
    keep = []
    s = _settings_struct(rs, dev, M, keep)
    s.flags = int(flags)                                          # SG_FLAG_* promises of the caller (tests: direct binning)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    cap = capacity or max(8 * P + T, 1 << 16)
    c = lambda t: None if t is None else t.contiguous()
    means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp = map(
        c, (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp))
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        while True:
            L = _lib.layout(P, W, H, cap)
            geom = torch.zeros(L.geom_bytes, dtype=torch.uint8, device=dev)
            binning = torch.zeros(L.bin_bytes, dtype=torch.uint8, device=dev)
            img = torch.zeros(L.img_bytes, dtype=torch.uint8, device=dev)
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            nr = C.c_int64(0)
            _lib.check(lib.sg_rasterize_forward(
                C.byref(s), P, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
                _ptr(rotations), _ptr(cov3D_precomp), _ptr(geom), _ptr(binning), cap, _ptr(img), _ptr(color),
                _ptr(radii), 1, C.byref(nr), stream), "forward")
            R = int(nr.value)
            if R < 0:                                            # NUM_RENDERED_LONG_LIST: a promise of `flags` did not hold
                return dict(color=color, radii=radii, R=R, capacity=cap, layout=L)
            if R <= cap:
                break
            cap = R + 1024

    def view(buf, off, n, dtype):
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype)

    # one 64-B record per Gaussian: recA, recB, recC, padding (SG_GEOM_REC_BYTES)
    assert L.geom_recB - L.geom_recA == 16 and L.geom_recC - L.geom_recA == 32
    rec = view(geom, L.geom_recA, P * 16, torch.float32).view(P, 16)
    recA, recB, recC = rec[:, 0:4], rec[:, 4:8], rec[:, 8:12]
    recCi = view(geom, L.geom_recA, P * 16, torch.int32).view(P, 16)[:, 8:12]
    out = dict(
        color=color, radii=radii, R=R, capacity=cap, geom=geom, binning=binning, img=img, layout=L,
        xy=recA[:, :2], conic_opacity=torch.stack([recA[:, 2], recA[:, 3], recB[:, 0], recB[:, 1]], 1),
        rgb=torch.stack([recB[:, 2], recB[:, 3], recC[:, 0]], 1),
        depths=view(geom, L.geom_depth, P, torch.float32),
        clamp_bits=view(geom, L.geom_flags, P, torch.int32),
        goff=recCi[:, 1], rect_min=recCi[:, 2], rect_wh=recCi[:, 3],
        point_list=view(binning, L.bin_point_list, R, torch.int32),
        point_keys=view(binning, L.bin_point_keys, R, torch.int64),
        ranges=view(binning, L.bin_ranges, T * 2, torch.int32).view(T, 2),
        final_T=view(img, L.img_final_T, H * W, torch.float32).view(H, W),
        n_contrib=view(img, L.img_n_contrib, H * W, torch.int32).view(H, W),
    )
    return out
