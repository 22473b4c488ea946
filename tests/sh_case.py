"""Scenes that make the rasterizer evaluate its SH colour at the 512 directions of golden G9 (tests/golden/sh_golden.npz, generated
from the reference's own spherical_harmonics.py by tests/golden/gen_sh_golden.py).

The preprocess computes dir = normalize(mean - campos) in WORLD space and colours only Gaussians that
land on the image, so the directions are split by dominant axis into six groups and each group gets a camera at the origin that
looks along that axis (tan(fov/2) = 2: |other / dominant| <= sqrt(2) stays on screen); Gaussian i sits at depth_i * dir_i.
"""
import os

import numpy as np

from sings_amd.camera import get_projection_matrix

G9 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sh_golden.npz"))
W = H = 64
TANFOV = 2.0


def groups():
    """-> list of dicts: idx (rows of G9), means3D, scales, rotations, opacities, viewmatrix, projmatrix, campos."""
    d = G9["dirs"].astype(np.float64)
    rs = np.random.RandomState(5)
    depth = rs.uniform(2.0, 4.0, d.shape[0])
    dom = np.abs(d).argmax(1)
    sign = np.sign(d[np.arange(d.shape[0]), dom])
    fov = 2.0 * np.arctan(TANFOV)
    P = get_projection_matrix(0.01, 100.0, fov, fov)
    out = []
    for ax in range(3):
        for sg in (1.0, -1.0):
            idx = np.nonzero((dom == ax) & (sign == sg))[0]
            z = np.zeros(3); z[ax] = sg                                  # the camera looks along +-axis
            x = np.zeros(3); x[(ax + 1) % 3] = 1.0
            y = np.cross(z, x)
            Rm = np.stack([x, y, z])                                     # p_view = Rm @ p
            view = np.eye(4, dtype=np.float32); view[:3, :3] = Rm.T      # row-vector convention: p_view = p @ view[:3,:3]
            proj = (view @ P.T).astype(np.float32)
            n = idx.size
            means = (d[idx] * depth[idx, None]).astype(np.float32)
            q = np.zeros((n, 4), np.float32); q[:, 0] = 1
            out.append(dict(idx=idx, means3D=means, scales=np.full((n, 3), 0.02, np.float32), rotations=q,
                            opacities=np.full((n, 1), 0.5, np.float32), viewmatrix=view, projmatrix=proj,
                            campos=np.zeros(3, np.float32), shs=np.ascontiguousarray(G9["sh"][idx])))
    assert sum(g["idx"].size for g in out) == d.shape[0]
    return out


def expected_rgb(deg, idx):
    """What the rasterizer stores for these Gaussians: eval_sh + 0.5, clamped at 0 (App. A.1 step 8), and the clamp flags."""
    v = G9[f"eval_deg{deg}"][idx] + np.float32(0.5)
    return np.maximum(v, 0), v < 0
