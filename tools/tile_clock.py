"""Per-tile wall clock of the forward composite on one avatar frame (GPU box; measurement build):
   bash tools/build_dbg.sh clk -DSG_TILE_CLOCK && SINGS_HIP_LIB=$PWD/build/dbg/libsings_hip_clk.so python tools/tile_clock.py"""
import ctypes as C, math, os, sys
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
A = joint_transforms(t(poses[0]), t(s["joints_rest"]), tuple(s["parents"])).reshape(J, 16).contiguous()
xyz, w, sc, op, sh = t(s["xyz_canon"]), t(s["lbs_weights"]), t(s["scales"]), t(s["opacities"]), t(s["shs"])
eng = SkinnedEngine(N, J, W, H, sh.shape[1], dev, capacity_pairs=16 * N + 65536)
eng.set_camera(rs); eng.set_frame(xyz, None, w, A, t(s["smpl_scale"]), t(s["transl"]))
for _ in range(5):
    R = eng.forward(sh, op, sc, sync_num_rendered=True)
T = ((W + 15) // 16) * ((H + 15) // 16)
lib = _lib.load()
buf = np.zeros((T, 4), np.uint64)
assert lib.sg_debug_tile_clock(buf.ctypes.data_as(C.c_void_p), T) == 0
t0, t1 = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
n = (buf[:, 2] & 0xffffffff).astype(np.int64); nw = ((buf[:, 2] >> 32) & 0xffff).astype(np.int64); nb = (buf[:, 2] >> 48).astype(np.int64)
cull = (buf[:, 3] & ((1 << 40) - 1)).astype(np.int64); livepx = (buf[:, 3] >> 40).astype(np.int64)
live = t1 > 0
k0 = t0[live].min()
dur = (t1 - t0) * 0.01                      # us (100 MHz)
print(f"R {R}  tiles {T}  kernel span {(t1[live].max() - k0) * 0.01:.1f} us; sum of tile times {dur[live].sum():.0f} us")
for lo, hi in ((0, 0), (1, 256), (257, 1024), (1025, 4096), (4097, 1 << 30)):
    m = live & (n >= lo) & (n <= hi)
    if m.any():
        print(f"  n {lo:>5}-{hi:<10}: {int(m.sum()):5d} tiles, mean {dur[m].mean():7.1f} us, max {dur[m].max():7.1f} us, windows {nw[m].sum():6d}, "
              f"batches {nb[m].sum():6d}, cull time {cull[m].sum() * 0.01:9.0f} us")
order = np.argsort(-dur)[:12]
print("longest tiles: (tile, n, windows, batches, us, of which cull us, start us, end us, pixels still live at the end)")
for i in order:
    print(f"   {i:5d} n={n[i]:6d} w={nw[i]:3d} b={nb[i]:3d} {dur[i]:7.1f} cull {cull[i] * 0.01:6.1f}  [{(t0[i] - k0) * 0.01:6.1f}, {(t1[i] - k0) * 0.01:6.1f}] live {livepx[i]}")
m = live & (nb >= 3)
print(f"tiles with >= 3 batches: {int(m.sum())}; of those never saturated (live pixels at the end): {int((m & (livepx > 0)).sum())}; "
      f"live-pixel histogram of those: {np.percentile(livepx[m & (livepx > 0)], [10, 50, 90]).tolist() if (m & (livepx > 0)).any() else []}")
late = np.argsort(-t1)[:8]
print("last to finish:", [(int(i), int(n[i]), round(float((t0[i] - k0) * 0.01), 1), round(float((t1[i] - k0) * 0.01), 1)) for i in late])
starts = np.sort((t0[live] - k0) * 0.01)
print("tile start times (us) percentiles 50/90/99/max:", [round(float(np.percentile(starts, p)), 1) for p in (50, 90, 99, 100)])
