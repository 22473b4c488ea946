"""HIP-graph replays with host synchronisations in between (GPU box): python tools/graph_gap.py

1. A minimal graph {hipMemsetAsync(buf, 0); buf += 1} built with torch.cuda.graph and a direct hipMemsetAsync call: every
   replay must leave buf == 1.  On ROCm 7.2 / gfx950 the replays AFTER the first host synchronisation do not (the memset
   node no longer takes effect as captured) -- the reason why libsings_hip zeroes its per-call counters with a kernel
   (sg_zero_async) instead of hipMemsetAsync.
2. The raster engine's captured step (RasterEngine.capture) replayed back to back / with synchronisations / with pauses:
   pair count, image and gradients must equal the directly launched step every time."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")

# ---- 1. memset node
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
buf = torch.zeros(1 << 16, dtype=torch.int32, device=dev)


def body():
    st = torch.cuda.current_stream(dev).cuda_stream
    err = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), 0, buf.numel() * 4, C.c_void_p(st))
    assert err == 0, err
    buf.add_(1)


side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
vals = []
for k in range(4):
    g.replay(); torch.cuda.synchronize()
    vals.append((int(buf.min()), int(buf.max())))
print("memset node: (min, max) of buf after each replay, a host synchronisation after every one:", vals,
      "OK" if all(v == (1, 1) for v in vals) else "<-- memset nodes do not survive replays after a synchronisation on this stack")

# ---- 2. the library's captured step
from sings_amd.engine import RasterEngine
from sings_amd.rasterizer import GaussianRasterizationSettings
from sings_amd.scene import synthetic_scene
s = synthetic_scene(50000, 512, 512, 3, 13)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rs = GaussianRasterizationSettings(image_height=s["H"], image_width=s["W"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
    scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), sh_degree=3, campos=t(s["campos"]),
    prefiltered=False, debug=False)
ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
dL = t(s["dL_dimage"])
eng = RasterEngine(50000, s["W"], s["H"], 16, dev, capacity_pairs=8 * 50000 + 65536)
eng.set_camera(rs)
eng.forward(*ins); eng.backward(*ins, dL); torch.cuda.synchronize()
ref_c, ref_g, ref_R = eng.color.clone(), eng.grad_flat.clone(), eng.num_rendered()
g2 = eng.capture(*ins, dL)
for mode in ("back-to-back", "sync between", "sleep between"):
    for k in range(3):
        g2.replay()
        if mode != "back-to-back":
            torch.cuda.synchronize()
        if mode == "sleep between":
            time.sleep(0.2)
    torch.cuda.synchronize()
    print("raster step,", mode, "R", eng.num_rendered(), "ref", ref_R, "image equal", bool(torch.equal(eng.color, ref_c)), "grads equal",
          bool(torch.equal(eng.grad_flat, ref_g)))
