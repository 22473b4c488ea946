"""Posed-frame driver: the callers' glue either side of the fused kernels.

 * ``joint_transforms_batch``: the SMPL(-H) kinematic chain for a CHUNK of frames in J batched 4x4 products
   (reference: ``forward_chunk`` -> ``smpl_template`` -> ``batch_rigid_transform``, sings_hybrid.py:474-531,
   body_model/smpl.py:462-513) -- without the per-step template work the reference discards
   (blend shapes, the J_regressor einsum over 110k vertices, full template skinning).
 * ``amass_to_smpl_pose``: the 156 -> 72 AMASS joint selection of sings/rec/defaults/constants.py:13-18.
 * ``animate_chunk``: forward-only rendering of a chunk of posed frames through the LBS-fused kernels, one
   fused call per frame (trainer counterpart: gs_trainer.py:664-728); posed means / quaternions never exist in HBM.
"""
import numpy as np
import torch

from .body import SMPL_PARENTS, rodrigues
from .renderer import get_render_pkg_fused

# sings/rec/defaults/constants.py:13-18: SMPL-H (52 joints, AMASS order) -> the 24 SMPL joints
AMASS_SMPLH_TO_SMPL_JOINTS = np.arange(0, 156).reshape((-1, 3))[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,
                                                                   17, 18, 19, 20, 21, 22, 37]].reshape(-1)


def amass_to_smpl_pose(poses156):
    """[F,156] AMASS SMPL-H axis-angle poses -> [F,72] SMPL poses (AnimDataset_opt.py:109-110)."""
    return poses156[..., AMASS_SMPLH_TO_SMPL_JOINTS]


def joint_transforms_batch(poses, joints_rest, parents=SMPL_PARENTS):
    """poses [B, J*3] axis-angle, joints_rest [J,3] -> A [B,J,4,4]; J-1 batched matmuls for the whole chunk."""
    B = poses.shape[0]
    J = joints_rest.shape[0]
    R = rodrigues(poses.reshape(B * J, 3)).view(B, J, 3, 3)
    rel = joints_rest.clone()
    par = torch.as_tensor(parents[1:], device=joints_rest.device, dtype=torch.long)
    rel[1:] = joints_rest[1:] - joints_rest[par]
    T = torch.zeros(B, J, 4, 4, dtype=poses.dtype, device=poses.device)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel[None]
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.bmm(chain[parents[i]], T[:, i]))
    G = torch.stack(chain, 1)
    jh = torch.cat([joints_rest, torch.zeros_like(joints_rest[:, :1])], 1)[None, :, :, None]
    corr = torch.matmul(G, jh)
    return G - torch.cat([torch.zeros(B, J, 4, 3, dtype=G.dtype, device=G.device), corr], -1)


@torch.no_grad()
def animate_chunk(canon, poses, joints_rest, A_t2cano, cameras, bg_color, transl=None, smpl_scale=None, ext_tfs=None,
                  parents=SMPL_PARENTS, chunk_size=16):
    """Renders frames ``poses[f]`` with camera dict ``cameras[f]`` (or one shared dict); yields (f, image[3,H,W]).

    canon: dict(xyz_canon, rotmat_canon|None, scales, opacity, shs, lbs_weights, active_sh_degree);
    transl [F,3] or None; ext_tfs: per-frame tuple (trans[F,3], rotmat[F,3,3], scale[F,1]) or None."""
    F = poses.shape[0]
    inv_cano = torch.inverse(A_t2cano)
    for c0 in range(0, F, chunk_size):
        A = joint_transforms_batch(poses[c0:c0 + chunk_size], joints_rest, parents) @ inv_cano[None]
        for i in range(A.shape[0]):
            f = c0 + i
            cam = cameras[f] if isinstance(cameras, (list, tuple)) else cameras
            ext = None if ext_tfs is None else (ext_tfs[0][f], ext_tfs[1][f], ext_tfs[2][f])
            pkg = get_render_pkg_fused(cam, canon, A[i], bg_color, smpl_scale=smpl_scale,
                                       transl=None if transl is None else transl[f], ext_tfs=ext)
            yield f, pkg["render"]
