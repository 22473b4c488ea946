"""G6: frozen outputs of the rasterizer restatement (oracle/raster_oracle.c) on small seeded scenes.

PARITY UNPINNED: these vectors are SELF-generated -- the upstream rasterizer (graphdeco-inria/diff-gaussian-rasterization,
un-pinned, install_all.sh:22) is not under /root/reference and cannot run here.  They freeze the restatement (so that an
accidental change of the oracle is caught, tests/test_oracle_raster.py) and give the GPU tests a committed fixture to compare
the HIP path with (tests/test_gpu_raster.py::test_against_frozen_oracle_vectors).  Re-generate only on purpose:
    python tests/golden/gen_raster_golden.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import raster_oracle as ro
from sings_amd.scene import synthetic_scene

CASES = {"a": (400, 96, 64, 3, 101), "b": (1500, 80, 112, 1, 102), "c": (60, 33, 17, 0, 103)}   # N, W, H, SH degree, seed
out = {}
for tag, (N, W, H, deg, seed) in CASES.items():
    s = synthetic_scene(N, W, H, deg, seed)
    o = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], W, H, s["tanfovx"], s["tanfovy"],
                   s["bg"], scales=s["scales"], rotations=s["rotations"], shs=s["shs"], sh_degree=deg)
    strict = o["margin"] >= 2e-5
    dL = s["dL_dimage"].copy(); dL[:, ~strict] = 0                 # gradients only through pixels far from a hard threshold
    g = ro.backward(o, dL)
    out[f"{tag}_case"] = np.array([N, W, H, deg, seed])
    for k in ("radii", "rect", "depths", "xy", "conic_opacity", "rgb", "point_list", "ranges", "color", "final_T", "n_contrib", "margin"):
        out[f"{tag}_{k}"] = o[k]
    out[f"{tag}_R"] = np.array(o["R"])
    out[f"{tag}_dL"] = dL
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drots", "dL_dopacity", "dL_dsh", "dL_dmeans2D"):
        if k in g:
            out[f"{tag}_{k}"] = g[k]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_golden.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith("a_")})
