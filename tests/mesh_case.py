"""A closed triangle mesh with vertex labels for the Laplacian regulariser tests (CPU oracle and GPU)."""
import numpy as np


def bumpy_sphere(nu=24, nv=16, seed=0):
    """closed triangle mesh (UV sphere with two pole vertices), radial noise; labels = 8 longitude sectors (0..7)."""
    rng = np.random.default_rng(seed)
    verts, lab = [[0, 0, 1.0]], [0]
    for j in range(1, nv):
        th = np.pi * j / nv
        for i in range(nu):
            ph = 2 * np.pi * i / nu
            r = 1.0 + 0.15 * rng.standard_normal()
            verts.append([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th)])
            lab.append(int(8 * i / nu))
    verts.append([0, 0, -1.0]); lab.append(7)
    idx = lambda j, i: 1 + (j - 1) * nu + (i % nu)
    faces = []
    for i in range(nu):
        faces.append([0, idx(1, i), idx(1, i + 1)])
        faces.append([len(verts) - 1, idx(nv - 1, i + 1), idx(nv - 1, i)])
    for j in range(1, nv - 1):
        for i in range(nu):
            faces.append([idx(j, i), idx(j + 1, i), idx(j + 1, i + 1)])
            faces.append([idx(j, i), idx(j + 1, i + 1), idx(j, i + 1)])
    return (np.asarray(verts, np.float32), np.asarray(faces, np.int64), np.asarray(lab, np.int64))
