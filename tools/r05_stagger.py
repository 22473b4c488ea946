"""GPU box: do two streams overlap better when their phases are staggered?  python tools/r05_stagger.py
cfg3 scene, 32 views per step on two streams, per-step join (the gradient fold is left out: rendering kernels only):
  aligned    A: [8][8]      B: [8][8]          (what bench.py does: both streams in the same phase all the time)
  staggered  A: [8][8]      B: [4][8][4]       (B's batches start half a batch off A's)
  staggered3 A: [8][8][8]   B: [4][8][8][4]    (48 views)
ms per view of each; the composite kernels are vector-issue bound, the per-Gaussian kernels memory bound: staggered, a stream's memory
phase can run beside the other's composite."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd.engine import RasterFramesEngine
from sings_amd.rasterizer import GaussianRasterizationSettings
from sings_amd.scene import synthetic_scene

dev = torch.device("cuda:0")
N, W, H, deg = 200000, 1920, 1080, 3
s = synthetic_scene(N, W, H, deg, 3)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
bg_t = t(s["bg"])


def camera(index):
    view = s["viewmatrix"].copy()
    view[3, 0] += 0.012 * (index % 8); view[3, 1] += 0.012 * ((index // 8) % 8)
    proj = (view @ P_T).astype(np.float32)
    campos = np.linalg.inv(view)[3, :3].astype(np.float32)
    return view, proj, campos


means3D, shs, opac, scales, rots = t(s["means3D"]), t(s["shs"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"])
dL = t(s["dL_dimage"])
base = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=bg_t, scale_modifier=1.0,
                                     viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), sh_degree=deg, campos=t(s["campos"]),
                                     prefiltered=False, debug=False)


def engine(K, first):
    e = RasterFramesEngine(N, W, H, shs.shape[1], K, dev, capacity_pairs=900000)
    cams = [camera(first + f) for f in range(K)]
    e.set_camera(base._replace(viewmatrix=t(np.stack([c[0] for c in cams])), projmatrix=t(np.stack([c[1] for c in cams])),
                               campos=t(np.stack([c[2] for c in cams]))), short_lists=True)
    return e, dL[None].expand(K, -1, -1, -1).contiguous()


def schedule(plan):
    """plan: list (one per stream) of lists of K"""
    streams = [torch.cuda.Stream(dev) for _ in plan]
    engs, v = [], 0
    for ks in plan:
        row = []
        for K in ks:
            row.append(engine(K, v)); v += K
        engs.append(row)
    views = v

    def step():
        cur = torch.cuda.current_stream(dev)
        for st in streams:
            st.wait_stream(cur)
        # issue round-robin over the streams so that no stream's launches wait for the host
        for i in range(max(len(r) for r in engs)):
            for st, row in zip(streams, engs):
                if i < len(row):
                    e, d = row[i]
                    with torch.cuda.stream(st):
                        e.forward(means3D, shs, opac, scales, rots)
                        e.backward(means3D, shs, opac, scales, rots, d)
        for st in streams:
            cur.wait_stream(st)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    return 1e3 * best / views, views


for name, plan in (("one stream [8][8][8][8]", [[8, 8, 8, 8]]),
                   ("aligned   [8][8] | [8][8]", [[8, 8], [8, 8]]),
                   ("staggered [8][8] | [4][8][4]", [[8, 8], [4, 8, 4]]),
                   ("staggered [8][8][8] | [4][8][8][4]", [[8, 8, 8], [4, 8, 8, 4]]),
                   ("staggered [16] | [8][8]... K16 | [4][8][4]", [[16], [4, 8, 4]]),
                   ("three     [8][8] | [4][8][4] | [2][8][6]", [[8, 8], [4, 8, 4], [2, 8, 6]]),
                   ("aligned   [8][8] | [8][8] (again)", [[8, 8], [8, 8]])):
    ms, views = schedule(plan)
    print(f"{name:48s} {views:3d} views  {ms:.4f} ms per view  {1e3 / ms:7.0f} views/s", flush=True)
    torch.cuda.empty_cache()
