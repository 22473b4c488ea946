"""Joint transforms of an SMPL-shaped kinematic tree, in torch (differentiable, runs on the MI355X).

Host-side producer of the ``A`` the fused LBS kernels consume.  Mirrors the math of the
reference's vendored SMPL chain -- ``batch_rodrigues`` (sings/rec/utils/body_model/smpl.py:415-446) and
``batch_rigid_transform`` (:462-513) -- WITHOUT the template work the reference throws away every step (blend
shapes, the J_regressor einsum over 110k vertices and the full template skinning): the rest
joints are computed once per shape and cached by the caller.  Gradients w.r.t. the pose flow through autograd, so
``dL/dA`` from ``sg_skinned_backward`` reaches ``body_pose`` / ``global_orient``.
"""
import torch

SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)   # smpl_layer.py:269-272


def rodrigues(rot_vecs):
    """[J,3] axis-angle -> [J,3,3]; angle = |theta + 1e-8| as the reference does."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    z = torch.zeros_like(d[:, 0:1])
    K = torch.cat([z, -d[:, 2:3], d[:, 1:2], d[:, 2:3], z, -d[:, 0:1], -d[:, 1:2], d[:, 0:1], z], 1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def joint_transforms(pose, joints_rest, parents=SMPL_PARENTS):
    """pose [J*3] or [J,3] axis-angle, joints_rest [J,3] -> A [J,4,4] (relative-to-rest joint transforms)."""
    J = joints_rest.shape[0]
    R = rodrigues(pose.reshape(J, 3))
    rel = joints_rest.clone()
    par = torch.as_tensor(parents[1:], device=joints_rest.device, dtype=torch.long)
    rel[1:] = joints_rest[1:] - joints_rest[par]
    T = torch.zeros(J, 4, 4, dtype=pose.dtype, device=pose.device)
    T[:, :3, :3] = R
    T[:, :3, 3] = rel
    T[:, 3, 3] = 1
    chain = [T[0]]
    for i in range(1, J):
        chain.append(chain[parents[i]] @ T[i])
    G = torch.stack(chain, 0)
    jh = torch.cat([joints_rest, torch.zeros_like(joints_rest[:, :1])], 1)[..., None]
    corr = G @ jh
    return G - torch.cat([torch.zeros(J, 4, 3, dtype=G.dtype, device=G.device), corr], -1)


def cano_to_pose(A_t2pose, A_t2cano):
    """A_cano2pose = A_t2pose @ inverse(A_t2cano)  (sings_hybrid.py:398-399, get_canonical_verts :578-596)."""
    return A_t2pose @ torch.inverse(A_t2cano)


def rotation_6d_to_matrix(d6):
    """sings/rec/utils/geometry/rotations.py:545-566 (Zhou et al. 6-D rotation): Gram-Schmidt of the two 3-vectors."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)
