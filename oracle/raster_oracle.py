"""ORACLE (test infrastructure only): ctypes front-end of oracle/libsgoracle.so.

CPU restatement of the rasterizer SinGS calls through ``diff_gaussian_rasterization``
(reference call sites sings/rec/renderer/gs_renderer_single.py:69-95).  PARITY UNPINNED:
the rasterizer source is an un-vendored, un-pinned dependency (install_all.sh:22); this
follows its published algorithm.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libsgoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "raster_core.inc.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.sgo_scan.restype = C.c_uint32
        _LIB.sgo_higher_msb.restype = C.c_uint32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=dt)


def higher_msb(n):
    return int(lib().sgo_higher_msb(C.c_uint32(n)))


def forward(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg,
            scales=None, rotations=None, shs=None, sh_degree=0, colors_precomp=None,
            cov3D_precomp=None, scale_modifier=1.0, dtype=np.float32, want_margin=True, border=2e-5):
    """Full forward.  Returns a dict with every intermediate of SURVEY.md App. A.1-A.3.  ``margin`` [H,W]: smallest relative
    distance of a pixel's hard-threshold decisions from their thresholds; ``flip`` [H,W]: bound on the colour change if the
    decisions closer than ``border`` go the other way (one splat's contribution each)."""
    L = lib()
    f64 = dtype == np.float64
    sfx = "_f64" if f64 else "_f32"
    real = C.c_double if f64 else C.c_float
    means3D = _c(means3D, dtype).reshape(-1, 3)
    P = means3D.shape[0]
    opacities = _c(opacities, dtype).reshape(-1)
    scales = _c(scales, dtype); rotations = _c(rotations, dtype)
    shs = _c(shs, dtype); colors_precomp = _c(colors_precomp, dtype); cov3D_precomp = _c(cov3D_precomp, dtype)
    view = _c(view, dtype).reshape(16); proj = _c(proj, dtype).reshape(16); campos = _c(campos, dtype).reshape(3)
    bg = _c(bg, dtype).reshape(3)
    M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    o = dict(P=P, W=W, H=H, gx=gx, gy=gy, M=M, D=sh_degree)
    o["radii"] = np.zeros(P, np.int32)
    o["xy"] = np.zeros((P, 2), dtype); o["depths"] = np.zeros(P, dtype)
    o["cov3D"] = np.zeros((P, 6), dtype); o["rgb"] = np.zeros((P, 3), dtype)
    o["conic_opacity"] = np.zeros((P, 4), dtype); o["clamped"] = np.zeros((P, 3), np.uint8)
    o["tiles_touched"] = np.zeros(P, np.uint32); o["rect"] = np.zeros((P, 4), np.int32)
    getattr(L, "sgo_preprocess" + sfx)(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales), real(scale_modifier),
        _p(rotations), _p(opacities), _p(shs), _p(colors_precomp), _p(cov3D_precomp),
        _p(view), _p(proj), _p(campos), C.c_int(W), C.c_int(H), real(tanfovx), real(tanfovy),
        _p(o["radii"]), _p(o["xy"]), _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]),
        _p(o["conic_opacity"]), _p(o["clamped"]), _p(o["tiles_touched"]), _p(o["rect"]))
    o["offsets"] = np.zeros(P, np.uint32)
    R = int(L.sgo_scan(C.c_int(P), _p(o["tiles_touched"]), _p(o["offsets"])))
    o["R"] = R
    xy32 = o["xy"].astype(np.float32); d32 = o["depths"].astype(np.float32)
    o["keys_unsorted"] = np.zeros(R, np.uint64); o["vals_unsorted"] = np.zeros(R, np.uint32)
    L.sgo_duplicate_with_keys(C.c_int(P), _p(xy32), _p(d32), _p(o["offsets"]), _p(o["radii"]),
                              C.c_int(gx), C.c_int(gy), _p(o["keys_unsorted"]), _p(o["vals_unsorted"]))
    o["keys"] = np.zeros(R, np.uint64); o["point_list"] = np.zeros(R, np.uint32)
    L.sgo_sort_pairs(C.c_uint32(R), _p(o["keys_unsorted"]), _p(o["vals_unsorted"]), _p(o["keys"]),
                     _p(o["point_list"]), C.c_int(32 + higher_msb(gx * gy)))
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    L.sgo_identify_ranges(C.c_uint32(R), _p(o["keys"]), C.c_int(gx * gy), _p(o["ranges"]))
    o["color"] = np.zeros((3, H, W), dtype); o["final_T"] = np.zeros((H, W), dtype)
    o["n_contrib"] = np.zeros((H, W), np.uint32)
    o["margin"] = np.ones((H, W), dtype) if want_margin else None
    o["flip"] = np.zeros((H, W), dtype) if want_margin else None
    cmax = float(max(np.abs(o["rgb"]).max() if P else 0.0, np.abs(bg).max()))
    getattr(L, "sgo_render_fwd" + sfx)(
        C.c_int(W), C.c_int(H), _p(o["ranges"]), _p(o["point_list"]), _p(o["xy"]), _p(o["rgb"]),
        _p(o["conic_opacity"]), _p(bg), _p(o["color"]), _p(o["final_T"]), _p(o["n_contrib"]), _p(o["margin"]),
        real(border), real(cmax), _p(o["flip"]))
    o["_in"] = dict(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations, shs=shs,
                    colors_precomp=colors_precomp, cov3D_precomp=cov3D_precomp, view=view, proj=proj,
                    campos=campos, bg=bg, tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier,
                    dtype=dtype)
    return o


def backward(o, dL_dpix):
    """Explicit backward (App. A.4 + A.5) of a forward() result.  Returns the grads dict."""
    L = lib()
    i = o["_in"]; dtype = i["dtype"]
    f64 = dtype == np.float64
    sfx = "_f64" if f64 else "_f32"
    real = C.c_double if f64 else C.c_float
    P, W, H, M, D = o["P"], o["W"], o["H"], o["M"], o["D"]
    dL_dpix = _c(dL_dpix, dtype).reshape(3, H, W)
    g = {}
    g["dL_dmean2D"] = np.zeros((P, 3), dtype); g["dL_dconic"] = np.zeros((P, 4), dtype)
    g["dL_dopacity"] = np.zeros((P, 1), dtype); g["dL_dcolor"] = np.zeros((P, 3), dtype)
    getattr(L, "sgo_render_bwd" + sfx)(
        C.c_int(P), C.c_int(W), C.c_int(H), _p(o["ranges"]), _p(o["point_list"]), _p(o["xy"]),
        _p(o["conic_opacity"]), _p(o["rgb"]), _p(i["bg"]), _p(o["final_T"]), _p(o["n_contrib"]),
        _p(dL_dpix), _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolor"]))
    g["dL_dmeans3D"] = np.zeros((P, 3), dtype); g["dL_dcov3D"] = np.zeros((P, 6), dtype)
    g["dL_dsh"] = np.zeros((P, max(M, 1), 3), dtype) if i["shs"] is not None else None
    g["dL_dscales"] = np.zeros((P, 3), dtype) if i["scales"] is not None else None
    g["dL_drots"] = np.zeros((P, 4), dtype) if i["rotations"] is not None else None
    getattr(L, "sgo_preprocess_bwd" + sfx)(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(i["means3D"]), _p(o["radii"]), _p(i["shs"]),
        _p(o["clamped"]), _p(i["scales"]), _p(i["rotations"]), real(i["scale_modifier"]),
        _p(o["cov3D"]), C.c_int(i["cov3D_precomp"] is not None), C.c_int(i["colors_precomp"] is not None),
        _p(i["view"]), _p(i["proj"]), _p(i["campos"]), C.c_int(W), C.c_int(H),
        real(i["tanfovx"]), real(i["tanfovy"]),
        _p(g["dL_dmean2D"]), _p(g["dL_dconic"]), _p(g["dL_dcolor"]),
        _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drots"]))
    return g
