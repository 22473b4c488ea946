"""ORACLE (test infrastructure only): a SECOND, independent restatement of the rasterizer SinGS calls through
``diff_gaussian_rasterization`` (call sites sings/rec/renderer/gs_renderer_single.py:69-95) -- vectorised torch,
gradients by torch.autograd.  SURVEY.md 8(c)(iii): the cross-check of the scalar C restatement
(oracle/raster_core.inc.c, explicit hand-derived backward).  PARITY UNPINNED like that file: the upstream source is an
un-vendored, un-pinned dependency (install_all.sh:22); both follow its published algorithm.

Independent means: written from the algorithm digest, not from the C file -- batched matrix algebra instead of scalar
code, tile lists from an argsort of the 64-bit keys, transmittance by cumprod, and NO hand-written chain rule: the
forward graph is differentiated by autograd.  The places where upstream's explicit backward is NOT the derivative of its
forward are expressed as small custom autograd nodes, each citing the digest:
  * ``_alpha_cap``      alpha = min(0.99, o G) with the gradient of the un-clamped product (App. A.4);
  * ``_conic``          inverse of the 2x2 covariance whose backward divides by det^2 + 1e-7 (App. A.5);
  * ``_clamp_frozen``   the field-of-view clamp of t.x / t.z: a clamped coordinate is a CONSTANT in backward (A.5:
                        "zero x/y contribution where the fov clamp was active"; t.z gets no term through it either);
  * the scale gradient  dL/dscale is reported w.r.t. the MODIFIED scale mod * s (upstream's computeCov3D backward has no
                        factor mod; identical for mod = 1, the only value the reference differentiates at).
tests/test_oracle_raster_twin.py compares the two restatements: integer outcomes (radii, tile rectangles, keys, sorted
lists, ranges, n_contrib) in fp32 exactly, images and every gradient in fp64 to ~1e-9 relative.
"""
import math

import numpy as np
import torch

BLOCK = 16

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


class _AlphaCap(torch.autograd.Function):
    """min(0.99, x) in forward, identity in backward."""

    @staticmethod
    def forward(ctx, x):
        return torch.clamp(x, max=0.99)

    @staticmethod
    def backward(ctx, g):
        return g


class _Conic(torch.autograd.Function):
    """(a, b, c) -> (c, -b, a) / det; backward with upstream's 1 / (det^2 + 1e-7) and its half-weight convention for the
    off-diagonal entry folded in (the total derivative w.r.t. b; App. A.5)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c)
        inv = 1.0 / det
        return c * inv, -b * inv, a * inv

    @staticmethod
    def backward(ctx, gx, gy, gz):
        # gx, gy, gz: true gradients w.r.t. conic (x, y, z).  Upstream carries dL/dconic.y at HALF weight through its render
        # backward and doubles it here; in terms of the true gradient gy that is dLy = gy / 2 in its formulas.
        a, b, c = ctx.saved_tensors
        det = a * c - b * b
        d2 = 1.0 / (det * det + 0.0000001)
        hy = 0.5 * gy
        da = d2 * (-c * c * gx + 2.0 * b * c * hy + (det - a * c) * gz)
        dc = d2 * (-a * a * gz + 2.0 * a * b * hy + (det - a * c) * gx)
        db = d2 * 2.0 * (b * c * gx - (det + 2.0 * b * b) * hy + a * b * gz)
        return da, db, dc


def _clamp_frozen(coord, tz, lim):
    """t.x := clamp(t.x / t.z, +-lim) * t.z.  Inside the limits this is t.x itself; outside it is a CONSTANT in backward
    (neither t.x nor t.z receives a gradient through it)."""
    ratio = coord / tz
    inside = (ratio >= -lim) & (ratio <= lim)
    return torch.where(inside, coord, (torch.clamp(ratio, -lim, lim) * tz).detach())


def _sh_colour(deg, dirs, sh):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = _C0 * sh[:, 0]
    if deg > 0:
        res = res - _C1 * y * sh[:, 1] + _C1 * z * sh[:, 2] - _C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _C2[0] * xy * sh[:, 4] + _C2[1] * yz * sh[:, 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + _C2[3] * xz * sh[:, 7] + _C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + _C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + _C3[1] * xy * z * sh[:, 10]
                       + _C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11] + _C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + _C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + _C3[5] * z * (xx - yy) * sh[:, 14]
                       + _C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res + 0.5


def rasterize(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg, scales, rotations, shs, sh_degree,
              scale_modifier=1.0, dtype=torch.float64):
    """Forward of the whole rasterizer as ONE autograd graph.  Inputs: numpy arrays or tensors (leaf tensors with
    requires_grad keep their identity so that ``.grad`` is filled by ``out['color'].backward(...)``).
    Returns a dict: color [3,H,W], final_T, n_contrib, radii, rect, tiles_touched, depths, keys, point_list, ranges, R, and
    ``ndc`` -- the retained [P,2] NDC means whose gradient is the op's ``means2D.grad[:, :2]`` (App. A.4)."""
    T = lambda a: a.to(dtype) if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), dtype=dtype)
    p, op, s, q, sh = T(means3D), T(opacities).reshape(-1), T(scales), T(rotations), T(shs)
    Vm, Pm, cam, bgc = T(view).reshape(4, 4), T(proj).reshape(4, 4), T(campos).reshape(3), T(bg).reshape(3)
    P = p.shape[0]
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)

    # ---- A.1 preprocess (row-vector convention: [p, 1] @ M with M the row-major [4,4] tensor)
    # (element-wise, left to right, so that an fp32 run rounds exactly like m0 x + m4 y + m8 z + m12 of the digest)
    xf = lambda M, k: M[0, k] * p[:, 0] + M[1, k] * p[:, 1] + M[2, k] * p[:, 2] + M[3, k]
    t = torch.stack([xf(Vm, 0), xf(Vm, 1), xf(Vm, 2)], 1)
    ph = torch.stack([xf(Pm, 0), xf(Pm, 1), xf(Pm, 2), xf(Pm, 3)], 1)
    in_front = t[:, 2] > 0.2
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * pw[:, None]
    if ndc.requires_grad:
        ndc.retain_grad()
    # 3-D covariance R S^2 R^T, un-normalised quaternion; gradient w.r.t. the modified scale
    s_eff = s + (scale_modifier - 1.0) * s.detach()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    RS = R * s_eff[:, None, :]
    Sigma = RS @ RS.transpose(1, 2)
    # EWA: cov2D = J W Sigma W^T J^T + 0.3 I
    tz_safe = torch.where(in_front, t[:, 2], torch.ones_like(t[:, 2]))
    tx = _clamp_frozen(t[:, 0], tz_safe, 1.3 * tanfovx)
    ty = _clamp_frozen(t[:, 1], tz_safe, 1.3 * tanfovy)
    zero = torch.zeros_like(tx)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], 1).reshape(P, 2, 3)
    Wv = Vm[:3, :3].t()                                   # rotation part: camera = Wv @ world
    M2 = J @ Wv
    cov = M2 @ Sigma @ M2.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    one = torch.ones_like(a)
    cx, cy, cz = _Conic.apply(torch.where(ok, a, one), torch.where(ok, b, zero), torch.where(ok, c, one))
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.clamp(lam, min=0.0))).detach()
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    mr = radius.to(torch.int64)
    mrf = mr.to(dtype)
    pd = pix.detach()
    trunc = lambda v: v.to(torch.int64)                   # C truncation toward zero
    x0 = trunc((pd[:, 0] - mrf) / BLOCK).clamp(0, gx); y0 = trunc((pd[:, 1] - mrf) / BLOCK).clamp(0, gy)
    x1 = trunc((pd[:, 0] + mrf + (BLOCK - 1)) / BLOCK).clamp(0, gx); y1 = trunc((pd[:, 1] + mrf + (BLOCK - 1)) / BLOCK).clamp(0, gy)
    tiles = (x1 - x0) * (y1 - y0)
    ok = ok & (tiles > 0)
    # colour
    d = p - cam[None]
    d = d / torch.sqrt((d * d).sum(1, keepdim=True))
    rgb = torch.clamp(_sh_colour(sh_degree, d, sh), min=0.0)

    radii = torch.where(ok, mr, torch.zeros_like(mr))
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    depth32 = t[:, 2].detach().to(torch.float32).numpy()

    # ---- A.2 keys in Gaussian order (y outer, x inner), stable sort
    keys, vals = [], []
    x0n, y0n, x1n, y1n, okn = x0.numpy(), y0.numpy(), x1.numpy(), y1.numpy(), ok.numpy()
    for i in range(P):
        if not okn[i]:
            continue
        dbits = int(np.frombuffer(np.float32(depth32[i]).tobytes(), np.uint32)[0])
        for yy in range(y0n[i], y1n[i]):
            for xx in range(x0n[i], x1n[i]):
                keys.append(((yy * gx + xx) << 32) | dbits); vals.append(i)
    keys = np.array(keys, np.uint64); vals = np.array(vals, np.int64)
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    for tl in np.unique(tile_of):
        w = np.nonzero(tile_of == tl)[0]
        ranges[tl] = (w[0], w[-1] + 1)

    # ---- A.3 compositing, one tile at a time, vectorised over (entry, pixel)
    color = torch.zeros((3, H, W), dtype=dtype) + bgc[:, None, None]
    final_T = torch.ones((H, W), dtype=dtype)
    n_contrib = np.zeros((H, W), np.uint32)
    conic = torch.stack([cx, cy, cz], 1)
    planes = []
    for tl in range(gx * gy):
        lo, hi = int(ranges[tl, 0]), int(ranges[tl, 1])
        X0, Y0 = (tl % gx) * BLOCK, (tl // gx) * BLOCK
        ys, xs = torch.meshgrid(torch.arange(Y0, min(Y0 + BLOCK, H)), torch.arange(X0, min(X0 + BLOCK, W)), indexing="ij")
        pxs, pys = xs.reshape(-1).to(dtype), ys.reshape(-1).to(dtype)
        if hi <= lo:
            continue
        ids = torch.from_numpy(vals[lo:hi])
        dx = pix[ids, 0:1] - pxs[None]; dy = pix[ids, 1:2] - pys[None]                     # [n, npix]
        cn = conic[ids]
        power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
        G = torch.exp(power)
        alpha = _AlphaCap.apply(op[ids][:, None] * G)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
        keep = 1.0 - a_eff
        T_before = torch.cat([torch.ones_like(keep[:1]), torch.cumprod(keep, 0)[:-1]], 0)
        term = valid & ((T_before * keep).detach() < 0.0001)
        n = hi - lo
        idx = torch.arange(n)[:, None].expand_as(term)
        first_term = torch.where(term, idx, torch.full_like(idx, n)).min(0).values         # per pixel
        blended = valid & (idx < first_term[None])
        wgt = torch.where(blended, alpha * T_before, torch.zeros_like(alpha))
        C = (rgb[ids].t()[:, :, None] * wgt[None]).sum(1)                                   # [3, npix]
        Tfin = torch.where(blended, keep, torch.ones_like(keep)).prod(0)
        last = torch.where(blended, idx + 1, torch.zeros_like(idx)).max(0).values
        h, w_ = ys.shape
        planes.append((Y0, X0, h, w_, C + Tfin[None] * bgc[:, None], Tfin, last))
    for Y0, X0, h, w_, col, Tf, last in planes:
        color[:, Y0:Y0 + h, X0:X0 + w_] = col.reshape(3, h, w_)
        final_T[Y0:Y0 + h, X0:X0 + w_] = Tf.reshape(h, w_)
        n_contrib[Y0:Y0 + h, X0:X0 + w_] = last.reshape(h, w_).numpy().astype(np.uint32)
    return dict(color=color, final_T=final_T.detach(), n_contrib=n_contrib, radii=radii.numpy().astype(np.int32),
                rect=torch.stack([x0, y0, x1, y1], 1).numpy().astype(np.int32), tiles_touched=tiles.numpy().astype(np.uint32),
                depths=t[:, 2].detach(), keys=keys, point_list=vals.astype(np.uint32), ranges=ranges, R=int(len(keys)),
                ndc=ndc, xy=pix.detach(), rgb=rgb.detach(), conic=conic.detach(), visible=ok.numpy())


def forward_backward(s, dL_dimage, dtype=torch.float64, scale_modifier=1.0):
    """Scene dict of sings_amd.scene.synthetic_scene -> (forward dict, gradients dict with the C oracle's key names)."""
    leaf = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype).clone().requires_grad_(True)
    m, o, sc, rt, sh = leaf(s["means3D"]), leaf(s["opacities"]), leaf(s["scales"]), leaf(s["rotations"]), leaf(s["shs"])
    out = rasterize(m, o, s["viewmatrix"], s["projmatrix"], s["campos"], s["W"], s["H"], s["tanfovx"], s["tanfovy"], s["bg"],
                    sc, rt, sh, s["sh_degree"], scale_modifier=scale_modifier, dtype=dtype)
    out["color"].backward(torch.as_tensor(np.asarray(dL_dimage), dtype=dtype))
    z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
    g = dict(dL_dmeans3D=z(m).numpy(), dL_dopacity=z(o).numpy().reshape(-1, 1), dL_dscales=z(sc).numpy(), dL_drots=z(rt).numpy(),
             dL_dsh=z(sh).numpy(), dL_dmean2D=(torch.zeros_like(out["ndc"]) if out["ndc"].grad is None else out["ndc"].grad).numpy())
    out["color"] = out["color"].detach().numpy()
    return out, g
