"""Forward-only animation throughput (anim_avatar.py -> animate_chunk) on the GPU box:
python tools/anim_time.py  -- 150k-Gaussian avatar, 120 AMASS frames, 512x896; 1 frame at a time vs frames in flight."""
import math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sings_amd.body import joint_transforms
from sings_amd.posed import FrameAnimator, animate_chunk
from sings_amd.scene import avatar_scene
dev = torch.device("cuda:0")
s = avatar_scene(N=150000, J=52)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cam = s["cam"]
data = dict(fovx=cam["fovx"], fovy=cam["fovy"], image_height=s["H"], image_width=s["W"],
            world_view_transform=t(cam["world_view_transform"]), full_proj_transform=t(cam["full_proj_transform"]),
            camera_center=t(cam["camera_center"]))
poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]
F, J = poses72.shape[0], s["J"]
poses = np.zeros((F, J * 3), np.float32); poses[:, :72] = poses72; poses[:, :3] = 0
poses = t(poses); jr = t(s["joints_rest"])
canon = dict(xyz_canon=t(s["xyz_canon"]), rotmat_canon=None, scales=t(s["scales"]), opacity=t(s["opacities"]), shs=t(s["shs"]),
             lbs_weights=t(s["lbs_weights"]), active_sh_degree=0)
A_cano = joint_transforms(torch.zeros(J * 3, device=dev), jr, tuple(s["parents"]))
transl = t(np.tile(s["transl"], (F, 1)))
bg = t(s["bg"])
scale = t(s["smpl_scale"])
for streams, chunk, K in ((1, 16, 1), (2, 16, 1), (3, 16, 1), (4, 16, 1), (6, 16, 1), (8, 16, 1),
                         (1, 16, 8), (2, 16, 8), (3, 16, 8), (1, 16, 16), (2, 16, 16), (3, 16, 16)):
    anim = FrameAnimator(canon, streams) if streams > 1 or K > 1 else None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 0
        for f, img in animate_chunk(canon, poses, jr, A_cano, data, bg, transl=transl, smpl_scale=scale,
                                    parents=tuple(s["parents"]), chunk_size=chunk, animator=anim, frames_per_launch=K):
            n += 1
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"streams={streams} chunk={chunk} frames per launch={K}: {n / el:8.1f} frames/s ({el / n * 1e3:.3f} ms per frame, {n} frames, joint transforms included)")
