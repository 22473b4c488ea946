"""Times the regulariser kernels on the GPU box: python tools/reg_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd.regularizers import GaussiansEdgeLoss, L2Norm, RegionLaplacianLoss_v2, mesh_edge_loss, knn_mean_edge
from sings_amd.scene import avatar_scene
dev = torch.device("cuda:0")


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


SIZES = [int(a) for a in sys.argv[1:]] or [150000, 500000]
for N in SIZES:
    s = avatar_scene(N=N, J=52)
    x = torch.from_numpy(s["xyz_canon"]).to(dev)
    sc = torch.from_numpy(s["scales"]).to(dev).requires_grad_(True)
    op = torch.from_numpy(s["opacities"]).to(dev).requires_grad_(True)
    off = (0.002 * torch.randn_like(x)).requires_grad_(True)
    print(f"N={N}: knn mean edge (K=9) {timeit(lambda: knn_mean_edge(x)):8.1f} us | GaussiansEdgeLoss fwd+grad "
          f"{timeit(lambda: GaussiansEdgeLoss()({'xyz_canon': x, 'scales': sc})):8.1f} us | L2Norm fwd+grad "
          f"{timeit(lambda: L2Norm()({'xyz_offsets': off, 'scales': sc, 'opacity': op})):8.1f} us")
# mesh-sized graph: a 166 x 166 triangulated sheet = 27 556 vertices (SMPL upsampled once), 15 label stripes
n = 166
gx, gy = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
vid = gx * n + gy
E = np.concatenate([np.stack([vid[:-1].ravel(), vid[1:].ravel()], 1), np.stack([vid[:, :-1].ravel(), vid[:, 1:].ravel()], 1),
                    np.stack([vid[:-1, :-1].ravel(), vid[1:, 1:].ravel()], 1)])
verts = torch.from_numpy(np.stack([gx.ravel() * 0.01, gy.ravel() * 0.01, np.zeros(n * n)], 1).astype(np.float32)).to(dev)
labels = torch.from_numpy((gy.ravel() * 15 // n).astype(np.int64))
mod = RegionLaplacianLoss_v2(verts, E, labels, region_weights=np.ones(15))
xv = (verts + 0.001 * torch.randn_like(verts)).requires_grad_(True)
print(f"V={n*n}, E={len(E)}: RegionLaplacianLoss_v2 fwd+grad {timeit(lambda: mod(xv)):8.1f} us | mesh_edge_loss fwd+grad "
      f"{timeit(lambda: mesh_edge_loss(xv, E)):8.1f} us")
# CPU baseline on a bounded sample: brute-force torch k-NN (what knn_points does) on 20 000 points
xs = x[:20000].cpu()
t0 = time.perf_counter(); d = torch.cdist(xs, xs); v = torch.topk(d, 9, dim=1, largest=False).values; t1 = time.perf_counter() - t0
print(f"CPU torch brute-force kNN, 20 000 points, {torch.get_num_threads()} threads: {t1*1e3:.1f} ms (O(N^2): x56 at 150 k)")
