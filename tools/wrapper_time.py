"""Drop-in autograd surface vs the pre-allocated engine (GPU box): python tools/wrapper_time.py -- cfg3, fwd+bwd per view."""
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
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f"autograd surface, default 'sync' (early pair count through a mapped host word):   {timeit():.3f} ms per view")
# the round-1 / round-2 synchronous check for comparison: debug = False, but no early-count word -> stream synchronisation
_arm = rz._arm_early_count
rz._arm_early_count = lambda s, dev: None
print(f"autograd surface, 'sync' by stream synchronisation (round-1 behaviour):             {timeit():.3f} ms per view")
rz._arm_early_count = _arm
rz.set_overflow_check("async")
print(f"autograd surface, 'async' (pair count looked at one call later; opt-in):            {timeit():.3f} ms per view")
rz.set_overflow_check("deferred")
print(f"autograd surface, 'deferred' (device-side accumulator, polled; opt-in):             {timeit():.3f} ms per view")
rz.set_overflow_check("sync")
