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
