"""Motion-sequence preparation the animation driver needs before any frame is rendered:
host-side, once per sequence, a handful of 3x3 products -- numpy by nature, like the camera producers.

  manual_alignment   sings/rec/datasets/motion_utils.py:10-26   per-source placement constants
  rebase_smpl        sings/rec/datasets/motion_utils.py:29-51   (call site AnimDataset_opt.py:118-120)

``rebase_smpl`` keeps the reference's behaviour to the letter, including what looks unfinished there: the re-based
global orientations are computed and then DROPPED -- ``poses`` comes back unchanged -- while the translations are
rotated by  R_target R_0^-1  (R_target = rotation by pi about x, R_0 = first frame's global orientation), shifted so
that the first frame sits at the origin, pushed 20 units along z, and returned with the shape ``[N, 3, 1]`` the
reference's matmul leaves them in (AnimDataset_opt.py:126 flattens them again).
Pinned by tests/golden/motion_golden.npz (G7: the reference function itself, tests/golden/gen_motion_golden.py).
"""
import numpy as np
import torch


def manual_alignment(motion_type, motion_name=None):
    """(translation [3], axis-angle rotation [3] in radians, scale) for a motion source."""
    deg = np.pi / 180.0
    table = {"AMASS": ((0, 0, 10), (90, 0, 0), 0.5), "custom": ((0, 0, 0), (-0.5, 0, 0), 1)}
    trans, rot_deg, scale = table.get(motion_type, ((0, 0, 0), (0, 0, 0), 0.5))
    return np.array(trans), np.array(rot_deg) * deg, scale


def _rotation_from_axis_angle(aa):
    """Rodrigues' formula for ONE axis-angle vector, float64 -> [3,3]."""
    aa = np.asarray(aa, np.float64)
    angle = float(np.linalg.norm(aa))
    if angle < 1e-12:
        return np.eye(3)
    k = aa / angle
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * (K @ K)


def rebase_smpl(poses, transl, init_global_orient=None, init_transl=None):
    """poses [N, 3 + ...] (axis-angle, global orientation first), transl [N,3] -> (poses UNCHANGED, transl [N,3,1] fp32)."""
    first = poses[0, :3].detach().cpu().numpy() if torch.is_tensor(poses) else np.asarray(poses)[0, :3]
    t = transl.detach().cpu().numpy() if torch.is_tensor(transl) else np.asarray(transl)
    # the reference evaluates this chain in fp32 torch: R_target @ inv(R_0) @ t, then the shifts
    R0 = _rotation_from_axis_angle(first).astype(np.float32)
    target = _rotation_from_axis_angle([np.pi, 0.0, 0.0]).astype(np.float32)
    M = target @ np.linalg.inv(R0).astype(np.float32)
    moved = (M[None] @ t.reshape(-1, 3, 1).astype(np.float32)).astype(np.float32)
    moved = moved - moved[0]
    moved[:, 2] += 20.0
    out = torch.from_numpy(moved)
    if torch.is_tensor(transl):
        out = out.to(transl.device)
    return poses, out
