"""Generates tests/golden/rot_cam_golden.npz by importing the reference's rotation conversions
(sings/rec/utils/geometry/rotations.py, torch only) and its orbit / static cameras (sings/rec/datasets/utils.py, whose
module-level ``import cv2`` and ``graphics`` import -- cv2 again -- are satisfied by empty placeholder modules, likewise loguru).

    python tests/golden/gen_rot_cam_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
lg = types.ModuleType("loguru"); lg.logger = None                  # (some reference modules import loguru's logger)
sys.modules.setdefault("loguru", lg)
from sings.rec.utils.geometry import rotations as rot             # noqa: E402
from sings.rec.datasets import utils as du                        # noqa: E402

rs = np.random.RandomState(21)
out = {}
aa = rs.normal(0, 1.2, (300, 3)).astype(np.float32)
aa[:4] = np.array([[0, 0, 0], [1e-8, 0, 0], [0, 3.1, 0], [1e-4, -1e-4, 2e-4]], np.float32)        # small-angle branch, near pi
d6 = rs.normal(0, 1, (300, 6)).astype(np.float32)
q = rs.normal(0, 1, (300, 4)).astype(np.float32)
T = torch.from_numpy
out.update(aa=aa, d6=d6, q=q,
           aa_to_q=rot.axis_angle_to_quaternion(T(aa)).numpy(), aa_to_m=rot.axis_angle_to_matrix(T(aa)).numpy(),
           aa_to_d6=rot.axis_angle_to_rotation_6d(T(aa)).numpy(), d6_to_aa=rot.rotation_6d_to_axis_angle(T(d6)).numpy(),
           q_to_aa=rot.quaternion_to_axis_angle(T(q)).numpy(), q_to_m=rot.quaternion_to_matrix(T(q)).numpy(),
           m_to_aa=rot.matrix_to_axis_angle(rot.rotation_6d_to_matrix(T(d6))).numpy(),
           m_to_d6=rot.matrix_to_rotation_6d(rot.rotation_6d_to_matrix(T(d6))).numpy(),
           q_std=rot.standardize_quaternion(T(q)).numpy())
cams = du.get_rotating_camera(img_size=(896, 512), fov=0.35, dist=4.5, device='cpu', nframes=7)
for k in ("world_view_transform", "full_proj_transform", "camera_center", "cam_int"):
    out["orbit_" + k] = np.stack([c[k].numpy() for c in cams])
st = du.get_static_camera(img_size=256, fov=0.4, device='cpu')
for k in ("world_view_transform", "full_proj_transform", "camera_center", "cam_int"):
    out["static_" + k] = st[k].numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rot_cam_golden.npz"), **out)
print("wrote rot_cam_golden.npz", len(out))
