"""Drop-in autograd surface (GaussianRasterizer through torch autograd, cfg3, fwd+bwd per view) in its overflow-check modes, each twice
(GPU box): python tools/wrapper_time.py  -- per view: wall time and the host's submission time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from sings_amd import rasterizer as rz
from sings_amd.scene import synthetic_scene
dev = torch.device("cuda:0")
s = synthetic_scene(200000, 1920, 1080, 3, 3)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rs = GaussianRasterizationSettings(image_height=s["H"], image_width=s["W"], tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
    scale_modifier=1.0, viewmatrix=t(s["viewmatrix"]), projmatrix=t(s["projmatrix"]), sh_degree=3, campos=t(s["campos"]),
    prefiltered=False, debug=False)
req = lambda a: t(a).requires_grad_(True)
m, op, sh, sc, rt = req(s["means3D"]), req(s["opacities"]), req(s["shs"]), req(s["scales"]), req(s["rotations"])
dL = t(s["dL_dimage"])
rast = GaussianRasterizer(rs)
def step():
    for x in (m, op, sh, sc, rt): x.grad = None
    m2 = torch.zeros_like(m, requires_grad=True)
    color, radii = rast(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rt)
    color.backward(dL)
def timeit(n=50):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3
for mode in ("sync", "async", "async", "deferred", "deferred", "sync"):
    rz.set_overflow_check(mode)
    a, b = timeit()
    print(f"{mode:9s} {a:.3f} ms per view (host submission {b:.3f} ms)  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
