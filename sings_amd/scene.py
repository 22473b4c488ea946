"""Seeded synthetic scenes S(N, W, H, deg, seed) of SURVEY.md section 8(d).

Numpy only (numpy.random.RandomState), cast to fp32; used by tests and bench.py so that
the CPU oracle and the MI355X path see identical inputs.
"""
import math

import numpy as np

from .camera import get_projection_matrix, focal2fov


def synthetic_scene(N, W, H, deg, seed, M=16):
    rs = np.random.RandomState(seed)
    fx = fy = 1.2 * W
    tanfovx = W / (2 * fx)
    tanfovy = H / (2 * fy)
    fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
    view = np.eye(4, dtype=np.float32)
    P = get_projection_matrix(0.01, 100.0, fovx, fovy)
    proj = (view @ P.T).astype(np.float32)
    campos = np.zeros(3, np.float32)
    z = rs.uniform(2, 10, N)
    u = rs.uniform(-1, 1, N); v = rs.uniform(-1, 1, N)
    x = u * 1.1 * z * tanfovx; y = v * 1.1 * z * tanfovy
    near = rs.rand(N) < 0.01
    z = np.where(near, rs.uniform(-1, 0.2, N), z)
    means = np.stack([x, y, z], 1).astype(np.float32)
    sigma_px = np.exp(rs.normal(math.log(2.0), 0.5, N))
    s = (sigma_px * np.abs(z) / fx)[:, None] * np.exp(rs.normal(0, 0.3, (N, 3)))
    scales = s.astype(np.float32)
    q = rs.normal(0, 1, (N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    nonunit = rs.rand(N) < 0.10
    q = np.where(nonunit[:, None], q * (1 + 0.01 * rs.normal(0, 1, (N, 1))), q)
    rots = q.astype(np.float32)
    opac = rs.uniform(0.05, 1.0, (N, 1)).astype(np.float32)
    shs = np.concatenate([rs.normal(0, 1, (N, 1, 3)), rs.normal(0, 0.15, (N, M - 1, 3))], 1).astype(np.float32)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    dL = rs.normal(0, 1, (3, H, W)).astype(np.float32)
    return dict(N=N, W=W, H=H, sh_degree=deg, tanfovx=tanfovx, tanfovy=tanfovy, fovx=fovx, fovy=fovy,
                viewmatrix=view, projmatrix=proj, campos=campos, means3D=means, scales=scales,
                rotations=rots, opacities=opac, shs=shs, bg=bg, dL_dimage=dL)


# ---------------------------------------------------------------------------------------------------
# cfg4 (BASELINE.json configs[3]): avatar-shaped scene -- ~150k canonical Gaussians on a capsule-limbed body
# proxy, J = 52 joints (SMPL body + 2 x 14 hand joints), posed by AMASS frames, camera of the shipped kit
# (fx = fy = 5000, 512 x 896, transl z ~ 10).
_SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
_SMPL_REST = np.array([
    [0, 0, 0], [0.07, -0.09, 0], [-0.07, -0.09, 0], [0, 0.11, -0.02], [0.10, -0.47, 0], [-0.10, -0.47, 0],
    [0, 0.25, 0], [0.09, -0.87, -0.03], [-0.09, -0.87, -0.03], [0, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0, 0.51, -0.02], [0.08, 0.42, -0.01], [-0.08, 0.42, -0.01], [0, 0.60, 0.03], [0.18, 0.45, -0.02], [-0.18, 0.45, -0.02],
    [0.43, 0.44, -0.04], [-0.43, 0.44, -0.04], [0.68, 0.45, -0.04], [-0.68, 0.45, -0.04], [0.77, 0.44, -0.05],
    [-0.77, 0.44, -0.05]], np.float32)


def body_tree(J=52):
    """Rest joints [J,3] and parents for a 24-joint SMPL-like body (+ 14 joints per hand when J = 52)."""
    parents = list(_SMPL_PARENTS); rest = [r for r in _SMPL_REST]
    if J > 24:
        per = (J - 24) // 2
        for side, wrist in ((1.0, 20), (-1.0, 21)):
            for f in range(per):
                chain_pos = f % 3
                parents.append(wrist if chain_pos == 0 else len(parents) - 1)
                base = rest[wrist] if chain_pos == 0 else rest[-1]
                rest.append(base + np.array([side * 0.03, 0.01 * ((f // 3) - 2) if chain_pos == 0 else 0.0, 0.0], np.float32))
    return np.stack(rest).astype(np.float32)[:J], parents[:J]


def avatar_scene(N=150000, J=52, W=512, H=896, seed=4, M=16, isotropic=True):
    from .camera import make_camera
    rs = np.random.RandomState(seed)
    rest, parents = body_tree(J)
    # SMPL-H meshes put ~22 % of their vertices in the two hands; the rest is spread over the body bones by length
    hand = rs.rand(N) < (0.22 if J > 24 else 0.0)
    blen = np.linalg.norm(rest[1:24] - rest[np.array(parents[1:24])], axis=1) + 0.05
    child = np.where(hand, rs.randint(24, max(J, 25), N), 1 + rs.choice(23, N, p=blen / blen.sum()))
    child = np.minimum(child, J - 1)
    par = np.array(parents)[child]
    t = rs.uniform(0, 1, (N, 1)).astype(np.float32)
    seg = rest[par] * (1 - t) + rest[child] * t
    radius = np.where(child < 24, 0.055, 0.008)[:, None]
    xyz = (seg + rs.normal(0, 1, (N, 3)) * radius).astype(np.float32)
    w = np.zeros((N, J), np.float32)
    w[np.arange(N), par] = 1 - t[:, 0]
    w[np.arange(N), child] += t[:, 0]
    extra = rs.randint(0, J, N)
    w[np.arange(N), extra] += 0.1 * rs.rand(N)
    w /= w.sum(1, keepdims=True)
    sc = np.exp(rs.normal(np.log(0.006), 0.3, (N, 1))).astype(np.float32)
    scales = np.repeat(sc, 3, 1) if isotropic else (sc * np.exp(rs.normal(0, 0.3, (N, 3)))).astype(np.float32)
    opac = rs.uniform(0.3, 1.0, (N, 1)).astype(np.float32)
    shs = np.concatenate([rs.normal(0, 1, (N, 1, 3)), rs.normal(0, 0.15, (N, M - 1, 3))], 1).astype(np.float32)
    cam = make_camera(np.eye(4, dtype=np.float32), 5000.0, 5000.0, W / 2, H / 2, W, H)
    return dict(N=N, J=J, W=W, H=H, xyz_canon=xyz, lbs_weights=w.astype(np.float32), scales=scales, opacities=opac, shs=shs,
                rotmat_canon=None, joints_rest=rest, parents=parents, cam=cam, smpl_scale=np.array([1.0], np.float32),
                transl=np.array([-0.04, 0.09, 10.06], np.float32), bg=np.array([1, 1, 1], np.float32),
                dL_dimage=rs.normal(0, 1, (3, H, W)).astype(np.float32))


def morton_order(xyz, bits=10):
    """Permutation that sorts points along a 3-D Morton (Z-order) curve over their bounding box.  Canonical Gaussians
    stored in this order make consecutive Gaussians neighbours in space -- the reference gets that for free from mesh
    subdivision; a randomly ordered cloud (e.g. after densification appends) loses it.  Spatial order is what the
    per-workgroup tile histogram of the preprocess, the tri-plane gathers and the record sums profit from."""
    x = np.asarray(xyz, dtype=np.float64)
    lo, hi = x.min(0), x.max(0)
    q = np.clip(((x - lo) / np.maximum(hi - lo, 1e-12) * ((1 << bits) - 1)).astype(np.uint64), 0, (1 << bits) - 1)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")
