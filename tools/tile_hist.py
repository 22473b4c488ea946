"""Tile-list length histogram of the avatar workload (GPU box): python tools/tile_hist.py"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sings_amd import _lib
from sings_amd.body import joint_transforms
from sings_amd.engine import SkinnedEngine
from sings_amd.rasterizer import GaussianRasterizationSettings
from sings_amd.scene import avatar_scene
dev = torch.device("cuda:0")
N = 150000
s = avatar_scene(N=N, J=52); W, H, J = s["W"], s["H"], s["J"]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
cam = s["cam"]
rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5), tanfovy=math.tan(cam["fovy"] * 0.5),
    bg=t(s["bg"]), scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]), projmatrix=t(cam["full_proj_transform"]), sh_degree=0,
    campos=t(cam["camera_center"]), prefiltered=False, debug=False)
poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]
poses = np.zeros((poses72.shape[0], J * 3), np.float32); poses[:, :72] = poses72; poses[:, :3] = 0
jr = t(s["joints_rest"])
A = joint_transforms(t(poses[0]), jr, tuple(s["parents"])).reshape(J, 16).contiguous()
xyz, w, sc, op, sh = t(s["xyz_canon"]), t(s["lbs_weights"]), t(s["scales"]), t(s["opacities"]), t(s["shs"])
eng = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=16 * N + 65536)
eng.set_camera(rs); eng.set_frame(xyz, None, w, A, t(s["smpl_scale"]), t(s["transl"]))
R = eng.forward(sh, op, sc, sync_num_rendered=True)
L = eng.L; T = ((W + 15) // 16) * ((H + 15) // 16)
rng = eng.binning[L.bin_ranges:L.bin_ranges + T * 8].view(torch.int32).view(T, 2).cpu().numpy()
n = (rng[:, 1] - rng[:, 0]).astype(np.int64)
nc = eng.img[L.img_n_contrib:L.img_n_contrib + W * H * 4].view(torch.int32).cpu().numpy().reshape(H, W)
print("R", R, "tiles", T, "nonempty", int((n > 0).sum()))
for lo, hi in ((1, 256), (257, 1024), (1025, 4096), (4097, 8192), (8193, 16384), (16385, 1 << 30)):
    m = (n >= lo) & (n <= hi); print(f"  len {lo:>6}-{hi:<10} tiles {int(m.sum()):5d}  pairs {int(n[m].sum()):9d}")
print("top lens", np.sort(n)[-10:])
# deepest contributor per tile vs list length (how much of a list the forward actually walks)
gx = (W + 15) // 16
mc = np.zeros(T, np.int64)
for ty in range((H + 15) // 16):
    for tx in range(gx):
        mc[ty * gx + tx] = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].max()
big = np.argsort(n)[-10:]
print("top tiles (len, deepest contributor):", [(int(n[i]), int(mc[i])) for i in big])
print("sum len", int(n.sum()), "sum deepest", int(mc.sum()))
