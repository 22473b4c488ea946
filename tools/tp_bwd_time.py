"""Times the tri-plane backward alone (GPU box): python tools/tp_bwd_time.py  -- avatar-shaped and uniform points."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd.decode import HexPlaneField
from sings_amd.scene import avatar_scene
dev = torch.device("cuda:0")
cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [64, 64, 64], 'multires': [1, 2, 4]}
torch.manual_seed(0)
f = HexPlaneField(cfg, bounds=1.2, device=dev)
s = avatar_scene(N=150000, J=52)
clouds = {"avatar": torch.from_numpy(s["xyz_canon"]).to(dev), "uniform": torch.rand(150000, 3, device=dev) * 2.2 - 1.1}
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, pts in clouds.items():
    if only and name != only:
        continue
    x = pts.clone().requires_grad_(True)
    feats = f(x)
    g = torch.randn_like(feats)
    def run():
        for p in f.parameters(): p.grad = None
        x.grad = None
        feats.backward(g, retain_graph=True)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    print(f"{name}: tri-plane backward {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
