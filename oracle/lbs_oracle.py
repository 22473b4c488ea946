"""ORACLE (test infrastructure only): torch-CPU restatement of the SMPL-LBS deformation block of
SinGS.  Every function cites the reference lines it follows; tests/golden/lbs_golden.npz pins it
against outputs of the reference's own code imported in the build container
(tests/golden/gen_lbs_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

All functions are plain differentiable torch, so autograd of this file is the gradient reference
for the fused HIP backward (LBS^T).
"""
import math

import torch

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]   # smpl_layer.py:269-272


# ----------------------------------------------------------------------------- rotations.py
def quaternion_to_matrix(q):
    """sings/rec/utils/geometry/rotations.py:38-66 (real part first, divides by |q|^2)."""
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    """rotations.py:87-95: sqrt(max(0,x)) with zero subgradient at x <= 0."""
    pos = x > 0
    return torch.where(pos, torch.sqrt(torch.where(pos, x, torch.ones_like(x))), torch.zeros_like(x))


def matrix_to_quaternion(m):
    """rotations.py:98-149: four candidate quaternions, floor 0.1 on the divisor, pick the candidate
    whose |component| estimate is largest (first maximum).  Input need not be orthonormal; output is
    then NOT a unit quaternion (SURVEY.md section 7 'Non-orthonormal rotations')."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(m.shape[:-2] + (9,)).unbind(-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(-1)
    return torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)


def quaternion_multiply(a, b):
    """rotations.py:372-407: Hamilton product, then flip sign so the real part is >= 0."""
    aw, ax, ay, az = a.unbind(-1); bw, bx, by, bz = b.unbind(-1)
    o = torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)
    return torch.where(o[..., 0:1] < 0, -o, o)


def rotation_6d_to_matrix(d6):
    """rotations.py:545-566: Gram-Schmidt; rows of the result are b1, b2, b3."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


# ----------------------------------------------------------------------------- body_model/smpl.py
def batch_rodrigues(rot_vecs):
    """sings/rec/utils/body_model/smpl.py:415-446: angle = |theta + 1e-8|, R = I + sin K + (1-cos) K^2."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=rot_vecs.dtype)[None] + s * K + (1 - c) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """smpl.py:462-513: kinematic chain G_i = G_parent(i) [R_i | j_i - j_parent]; A = G - [0 | G (j;0)]."""
    B, J = rot_mats.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=rot_mats.dtype)
    T[:, :, :3, :3] = rot_mats; T[:, :, :3, 3] = rel; T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[parents[i]], T[:, i]))
    G = torch.stack(chain, 1)
    posed = G[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros(B, J, 1, dtype=joints.dtype)], -1)[..., None]      # (j; 0)
    corr = torch.matmul(G, jh)                                                              # B,J,4,1
    A = G - torch.cat([torch.zeros(B, J, 4, 3, dtype=G.dtype), corr], -1)
    return posed, A


def smpl_lbs(betas, pose, v_template, shapedirs, J_regressor, parents, lbs_weights):
    """smpl.py:274-368 (`lbs`, pose blend shapes unused there: v_posed = v_shaped :339)."""
    B = max(betas.shape[0], pose.shape[0])
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)          # :391-412
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)                          # :371-388
    R = batch_rodrigues(pose.view(-1, 3)).view(B, -1, 3, 3)
    J_t, A = batch_rigid_transform(R, J, parents)
    verts, T = lbs_extra(A, v_shaped, lbs_weights)
    return verts, J_t, A, T


def lbs_extra(A, v, lbs_weights):
    """sings/rec/utils/body_model/lbs.py:59-74: T = W . A (N x J times J x 16), v' = (T [v;1])[:3].
    (rodrigues/pose_feature at :31-34 are dead code when disable_posedirs=True, the only mode used:
    sings_hybrid.py:57, gs_trainer.py:91-105.)"""
    B, J = A.shape[:2]
    W = lbs_weights[None].expand(B, -1, -1)
    T = torch.matmul(W, A.reshape(B, J, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v, torch.ones_like(v[..., :1])], -1)
    verts = torch.matmul(T, vh[..., None])[:, :, :3, 0]
    return verts, T


# ----------------------------------------------------------------------------- sings_hybrid.py:398-428
def deform_gaussians(xyz_canon, rotmat_canon, scales, lbs_weights, A_cano2pose, smpl_scale=None, transl=None,
                     ext_tfs=None):
    """The LBS block of SinGS.forward (sings/rec/models/sings_hybrid.py:400-428), one frame.
    xyz_canon [N,3], rotmat_canon [N,3,3], scales [N,3], lbs_weights [N,J], A_cano2pose [J,4,4],
    smpl_scale [1], transl [3], ext_tfs = (trans[3], rotmat[3,3], scale[1]) or None.
    Returns posed xyz [N,3], rotq [N,4] (real first, NOT normalised), scales [N,3], lbs_T [N,4,4]."""
    xyz, T = lbs_extra(A_cano2pose[None], xyz_canon[None], lbs_weights)          # :400-406
    xyz, T = xyz[0], T[0]
    if smpl_scale is not None:                                                   # :411-413
        xyz = xyz * smpl_scale[None]
        scales = scales * smpl_scale[None]
    if transl is not None:                                                       # :415-416
        xyz = xyz + transl[None]
    R_def = T[:, :3, :3] @ rotmat_canon                                          # :418
    q = matrix_to_quaternion(R_def)                                              # :419
    if ext_tfs is not None:                                                      # :421-428
        tr, rm, sc = ext_tfs
        xyz = (tr[..., None] + (sc[None] * (rm @ xyz[..., None]))).squeeze(-1)
        scales = sc * scales
        q = quaternion_multiply(matrix_to_quaternion(rm), q)
    return xyz, q, scales, T


# ----------------------------------------------------------------------------- synthetic SMPL-shaped model
def synthetic_body_model(seed=0, V=6890, J=24, nbetas=10):
    """Seeded stand-in for the licensed SMPL pickle: random template, shape dirs,
    row-normalised sparse-ish joint regressor and skinning weights."""
    import numpy as np
    rs = np.random.RandomState(seed)
    v_template = rs.normal(0, 0.3, (V, 3)).astype(np.float32)
    shapedirs = rs.normal(0, 0.01, (V, 3, nbetas)).astype(np.float32)
    Jr = rs.rand(J, V).astype(np.float32) ** 8
    Jr /= Jr.sum(1, keepdims=True)
    w = rs.rand(V, J).astype(np.float32) ** 6
    w[w < 0.05] = 0
    w[np.arange(V), rs.randint(0, J, V)] += 0.5
    w /= w.sum(1, keepdims=True)
    parents = SMPL_PARENTS if J == 24 else [-1] + [int(rs.randint(0, i)) for i in range(1, J)]
    return dict(v_template=v_template, shapedirs=shapedirs, J_regressor=Jr, lbs_weights=w.astype(np.float32),
                parents=np.array(parents, np.int64))
