"""Debug (GPU box): RasterEngine step replayed from a HIP graph with / without host synchronisation between replays."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd.engine import RasterEngine
from sings_amd.rasterizer import GaussianRasterizationSettings
from sings_amd.scene import synthetic_scene
dev = torch.device("cuda:0")
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
g = eng.capture(*ins, dL)
for mode in ("back-to-back", "sync between", "sleep between"):
    for k in range(3):
        g.replay()
        if mode == "sync between":
            torch.cuda.synchronize()
        if mode == "sleep between":
            torch.cuda.synchronize(); import time; time.sleep(0.2)
    torch.cuda.synchronize()
    print(mode, "R", eng.num_rendered(), "ref", ref_R, "image equal", bool(torch.equal(eng.color, ref_c)), "grads equal", bool(torch.equal(eng.grad_flat, ref_g)))
