"""ORACLE / CPU baseline (test infrastructure only): vectorised PyTorch-CPU "LBS + project".

BASELINE.md section 3: skinning (T = W.A, v' = T [v;1]), rotation compose + matrix_to_quaternion
(sings/rec/models/sings_hybrid.py:398-419, sings/rec/utils/body_model/lbs.py:59-74) followed by the forward
preprocess of the rasterizer (cull / 4x4 projection / 3-D covariance / EWA 2-D covariance / radius / SH colour,
SURVEY.md App. A.1).  It is the vectorised twin of oracle/raster_core.inc.c::sgo_preprocess (cross-checked in
tests/test_oracle_raster.py) and the thing bench.py times as `cpu_lbs_project`.
"""
import math

import torch

from . import lbs_oracle as lo

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            b += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
            if deg > 2:
                b += [C3[0] * y * (3.0 * xx - yy), C3[1] * xy * z, C3[2] * y * (4.0 * zz - xx - yy),
                      C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), C3[4] * x * (4.0 * zz - xx - yy),
                      C3[5] * z * (xx - yy), C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(b, 1)


def project(means3D, scales, rotations, opacities, shs, sh_degree, view, proj, campos, W, H, tanfovx, tanfovy,
            scale_modifier=1.0):
    """Vectorised App. A.1.  view / proj: [4,4] row-major tensors = column-major matrices.  Returns a dict with
    radii (int32), xy, depths, conic_opacity, rgb, cov3D, tiles_touched, rect."""
    p = means3D
    m = view.reshape(16); pm = proj.reshape(16)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    pv = [m[i] * x + m[4 + i] * y + m[8 + i] * z + m[12 + i] for i in range(3)]
    ph = [pm[i] * x + pm[4 + i] * y + pm[8 + i] * z + pm[12 + i] for i in range(4)]
    ok = pv[2] > 0.2
    pw = 1.0 / (ph[3] + 0.0000001)
    ppx, ppy = ph[0] * pw, ph[1] * pw
    # cov3D = R S^2 R^T with the un-normalised quaternion
    r, qx, qy, qz = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - r * qz), 2 * (qx * qz + r * qy),
         2 * (qx * qy + r * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - r * qx),
         2 * (qx * qz - r * qy), 2 * (qy * qz + r * qx), 1 - 2 * (qx * qx + qy * qy)]
    s = [scale_modifier * scales[:, k] for k in range(3)]
    M = [[s[k] * R[3 * i + k] for i in range(3)] for k in range(3)]
    S = [[M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j] for j in range(3)] for i in range(3)]
    c6 = [S[0][0], S[0][1], S[0][2], S[1][1], S[1][2], S[2][2]]
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    zs = torch.where(ok, pv[2], torch.ones_like(pv[2]))
    tx = torch.clamp(pv[0] / zs, -limx, limx) * zs
    ty = torch.clamp(pv[1] / zs, -limy, limy) * zs
    j00, j02 = fx / zs, -(fx * tx) / (zs * zs)
    j11, j12 = fy / zs, -(fy * ty) / (zs * zs)
    Mm0 = [j00 * m[0 + 4 * k] + j02 * m[2 + 4 * k] for k in range(3)]
    Mm1 = [j11 * m[1 + 4 * k] + j12 * m[2 + 4 * k] for k in range(3)]
    V = [[c6[0], c6[1], c6[2]], [c6[1], c6[3], c6[4]], [c6[2], c6[4], c6[5]]]
    t0 = [Mm0[0] * V[0][j] + Mm0[1] * V[1][j] + Mm0[2] * V[2][j] for j in range(3)]
    t1 = [Mm1[0] * V[0][j] + Mm1[1] * V[1][j] + Mm1[2] * V[2][j] for j in range(3)]
    a = t0[0] * Mm0[0] + t0[1] * Mm0[1] + t0[2] * Mm0[2] + 0.3
    b = t0[0] * Mm1[0] + t0[1] * Mm1[1] + t0[2] * Mm1[2]
    c = t1[0] * Mm1[0] + t1[1] * Mm1[1] + t1[2] * Mm1[2] + 0.3
    det = a * c - b * b
    ok = ok & (det != 0)
    det_inv = 1.0 / torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    pix = torch.stack([((ppx + 1.0) * W - 1.0) * 0.5, ((ppy + 1.0) * H - 1.0) * 0.5], 1)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    mr = radius.to(torch.int32)
    mrf = mr.to(p.dtype)
    trunc = lambda v: v.to(torch.int32)
    x0 = trunc((pix[:, 0] - mrf) / 16.0).clamp(0, gx); y0 = trunc((pix[:, 1] - mrf) / 16.0).clamp(0, gy)
    x1 = trunc((pix[:, 0] + mrf + 15.0) / 16.0).clamp(0, gx); y1 = trunc((pix[:, 1] + mrf + 15.0) / 16.0).clamp(0, gy)
    tt = (x1 - x0) * (y1 - y0)
    ok = ok & (tt > 0)
    d = p - campos[None]
    d = d / torch.sqrt((d * d).sum(1, keepdim=True))
    nc = (sh_degree + 1) ** 2
    basis = sh_basis(sh_degree, d)
    rgb = (basis[:, :, None] * shs[:, :nc, :]).sum(1) + 0.5
    clamped = rgb < 0
    rgb = torch.clamp(rgb, min=0.0)
    zero = torch.zeros_like
    okf = ok[:, None]
    return dict(radii=torch.where(ok, mr, zero(mr)), xy=torch.where(okf, pix, zero(pix)),
                depths=torch.where(ok, pv[2], zero(pv[2])),
                conic_opacity=torch.where(okf, torch.cat([conic, opacities.reshape(-1, 1)], 1), torch.zeros(len(p), 4, dtype=p.dtype)),
                rgb=torch.where(okf, rgb, zero(rgb)), clamped=clamped & okf, cov3D=torch.stack(c6, 1),
                tiles_touched=torch.where(ok, tt, zero(tt)), rect=torch.stack([x0, y0, x1, y1], 1))


def lbs_project(xyz_canon, rotmat_canon, scales, opacities, shs, sh_degree, lbs_weights, A, smpl_scale, transl, view, proj,
                campos, W, H, tanfovx, tanfovy):
    """The CPU baseline of BASELINE.md: deformation block + projection, no rasterisation."""
    xyz, q, sc, _ = lo.deform_gaussians(xyz_canon, rotmat_canon, scales, lbs_weights, A, smpl_scale=smpl_scale, transl=transl)
    return project(xyz, sc, q, opacities, shs, sh_degree, view, proj, campos, W, H, tanfovx, tanfovy)
