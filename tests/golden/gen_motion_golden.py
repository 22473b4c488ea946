"""Generates tests/golden/motion_golden.npz by IMPORTING the reference's own Python (read-only at /root/reference) in
the build container; only the vectors ship.

    python tests/golden/gen_motion_golden.py

G7 datasets/motion_utils.py:29-51  rebase_smpl on AMASS frames (the module imports pytorch3d.transforms, absent here; the
   six functions it uses are satisfied by the reference's OWN vendored copy of them, sings/rec/utils/geometry/rotations.py)
   and manual_alignment (:10-26).
G8 the batched deformation of SinGS.forward_chunk (sings_hybrid.py:512-553) evaluated with the reference's imported
   lbs_extra / rotations functions on B = 3 frames: xyz, scales, rotq with and without ext_tfs, isotropic and not.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from sings.rec.utils.geometry import rotations as rot            # noqa: E402
from sings.rec.utils.body_model import smpl as vsmpl             # noqa: E402

p3d = types.ModuleType("pytorch3d"); p3dt = types.ModuleType("pytorch3d.transforms")
for fn in ("axis_angle_to_quaternion", "quaternion_to_axis_angle", "quaternion_invert", "quaternion_multiply",
           "axis_angle_to_matrix", "matrix_to_axis_angle"):
    setattr(p3dt, fn, getattr(rot, fn))
sys.modules["pytorch3d"] = p3d; sys.modules["pytorch3d.transforms"] = p3dt
import importlib.util                                             # noqa: E402
spec = importlib.util.spec_from_file_location("ref_motion_utils", os.path.join(REF, "sings/rec/datasets/motion_utils.py"))
mu = importlib.util.module_from_spec(spec); spec.loader.exec_module(mu)

stub = types.ModuleType("smplx.lbs")
for fn in ("batch_rodrigues", "blend_shapes", "vertices2joints", "batch_rigid_transform"):
    setattr(stub, fn, getattr(vsmpl, fn))
sys.modules["smplx"] = types.ModuleType("smplx"); sys.modules["smplx.lbs"] = stub
from sings.rec.utils.body_model import lbs as rlbs               # noqa: E402

out = {}
T = lambda a: torch.from_numpy(np.asarray(a, np.float32))
rs = np.random.RandomState(77)

# ---- G7
amass = np.load(os.path.join(REF, "data/animation/AMASS/SFU/0008/0008_Walking002_poses.npz"))
sel = np.arange(0, 156).reshape((-1, 3))[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 37]].reshape(-1)
poses = torch.from_numpy(amass["poses"][200:800:40, sel])        # float64, as AnimDataset_opt.py:109 hands it over
transl = torch.from_numpy(amass["trans"][200:800:40])
out["g7_poses"] = poses.numpy(); out["g7_transl"] = transl.numpy()
p2, t2 = mu.rebase_smpl(poses.clone(), transl.clone())
out["g7_poses_out"] = p2.numpy(); out["g7_transl_out"] = t2.numpy()
for kind in ("AMASS", "custom", "other"):
    tr, ro_, sc = mu.manual_alignment(kind)
    out[f"g7_align_{kind}"] = np.concatenate([np.asarray(tr, np.float64), np.asarray(ro_, np.float64), [float(sc)]])

# ---- G8
B, N, J = 3, 500, 24
A = np.tile(np.eye(4, dtype=np.float32), (B, J, 1, 1))
for b in range(B):
    for j in range(J):
        A[b, j, :3, :3] = vsmpl.batch_rodrigues(T(rs.normal(0, 0.6, (1, 3)))).numpy()[0]
        A[b, j, :3, 3] = rs.normal(0, 0.2, 3)
w = rs.rand(N, J).astype(np.float32) ** 5; w /= w.sum(1, keepdims=True)
xyz_canon = rs.normal(0, 0.4, (N, 3)).astype(np.float32)
rot6d = rs.normal(size=(N, 6)).astype(np.float32)
scales = np.exp(rs.normal(-4, 0.3, (N, 3))).astype(np.float32)
smpl_scale = np.array([[1.07], [0.93], [1.0]], np.float32)       # [B,1]
transl_b = rs.normal(0, 1, (B, 3)).astype(np.float32)
ext_trans = rs.normal(0, 0.3, (B, 3)).astype(np.float32)
ext_rot = vsmpl.batch_rodrigues(T(rs.normal(0, 0.8, (B, 3)))).numpy()
ext_scale = np.array([[1.3], [0.8], [1.0]], np.float32)
for k, v in dict(A=A, w=w, xyz_canon=xyz_canon, rot6d=rot6d, scales=scales, smpl_scale=smpl_scale, transl=transl_b,
                 ext_trans=ext_trans, ext_rot=ext_rot, ext_scale=ext_scale).items():
    out["g8_" + k] = v
for iso in (False, True):
    for ext in (False, True):
        # sings_hybrid.py:512-553 with the imported pieces
        gs_xyz_canon = T(xyz_canon).unsqueeze(0).expand(B, -1, -1)
        rotmat_canon = torch.eye(3).unsqueeze(0).repeat(N, 1, 1) if iso else rot.rotation_6d_to_matrix(T(rot6d))
        gs_rotmat_canon = rotmat_canon.unsqueeze(0).expand(B, -1, -1, -1)
        gs_scales = T(scales).unsqueeze(0).expand(B, -1, -1)
        xyz_deformed, _, lbs_T, _, _ = rlbs.lbs_extra(T(A), gs_xyz_canon, posedirs=None, lbs_weights=T(w),
                                                      pose=torch.zeros(B, J * 3), disable_posedirs=True, pose2rot=True)
        xyz_deformed = xyz_deformed * T(smpl_scale).unsqueeze(-1)
        gs_scales = gs_scales * T(smpl_scale).unsqueeze(-1)
        xyz_deformed = xyz_deformed + T(transl_b).unsqueeze(1)
        gs_rotq = rot.matrix_to_quaternion(lbs_T[..., :3, :3] @ gs_rotmat_canon)
        if ext:
            trans, rotmat, scale = T(ext_trans), T(ext_rot), T(ext_scale)
            xyz_deformed = (trans[:, None, :] + (scale[:, None] * (rotmat[:, None, ...] @ xyz_deformed[..., None]).squeeze(-1)))
            gs_scales = scale[..., None] * gs_scales
            gs_rotq = rot.quaternion_multiply(rot.matrix_to_quaternion(rotmat)[:, None, :], gs_rotq)
        tag = f"g8_{'iso' if iso else 'aniso'}_{'ext' if ext else 'plain'}"
        out[tag + "_xyz"] = xyz_deformed.numpy(); out[tag + "_scales"] = gs_scales.numpy(); out[tag + "_rotq"] = gs_rotq.numpy()
        if not iso and not ext:
            out["g8_rotq_canon"] = rot.matrix_to_quaternion(rotmat_canon).numpy()

dst = os.path.join(ROOT, "tests", "golden", "motion_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst) // 1024, "KiB", {k: v.shape for k, v in out.items() if k.startswith("g7")})
