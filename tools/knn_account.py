"""What one point of the k-NN search costs (accounting build: tools/knn_account.sh).  Avatar cloud of the training bench."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sings_amd import _lib
from sings_amd.regularizers import knn_mean_edge
from sings_amd.scene import avatar_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
dev = torch.device("cuda:0")
xyz = torch.from_numpy(np.ascontiguousarray(avatar_scene(N=N, J=52)["xyz_canon"])).to(dev)
lib = _lib.load()
lib.sg_debug_knn_account.argtypes = [C.c_void_p]; lib.sg_debug_knn_account.restype = C.c_int
acct = torch.zeros((N, 8), dtype=torch.int32, device=dev)
assert lib.sg_debug_knn_account(C.c_void_p(acct.data_ptr())) == 0
knn_mean_edge(xyz, K=9); torch.cuda.synchronize()
assert lib.sg_debug_knn_account(C.c_void_p(0)) == 0
a = acct.cpu().numpy().astype(np.int64)
names = ["rows iterated", "rows searched", "cell look-ups", "candidates (distance evaluations)", "inserts", "rings"]
print(f"N = {N}, K = 9; points on the fine grid: {int(a[:, 6].sum())} ({100.0 * a[:, 6].mean():.1f} %)")
for q, nm in enumerate(names):
    v = a[:, q]
    print(f"  {nm:36s} mean {v.mean():8.1f}  p50 {np.percentile(v, 50):7.0f}  p90 {np.percentile(v, 90):7.0f}  p99 {np.percentile(v, 99):7.0f}  max {v.max():7d}  total {v.sum():.3e}")
busy = a[:, 7]
print(f"  candidates of a point's busiest lane   mean {busy.mean():8.1f}  p50 {np.percentile(busy, 50):7.0f}  p99 {np.percentile(busy, 99):7.0f}  max {busy.max():7d}")
# a wave = 16 consecutive sorted points x 4 lanes: its time follows its busiest lane.  (Order of `acct` is the original point
# order, not the sorted one, so this is the cloud-wide figure only.)
cand = a[:, 3].astype(np.float64)
print(f"  bytes gathered per point: candidates x 16 + look-ups x 8 = {(cand * 16 + a[:, 2] * 8).mean():.0f} B (mean); the 8 neighbours themselves are 128 B")
for lab, sel in (("coarse-grid points", a[:, 6] == 0), ("fine-grid points", a[:, 6] == 1)):
    if sel.any():
        print(f"  {lab}: {int(sel.sum())} points, candidates mean {cand[sel].mean():.0f} p99 {np.percentile(cand[sel], 99):.0f} max {cand[sel].max():.0f}; rings mean {a[sel, 5].mean():.2f} max {a[sel, 5].max()}; rows searched mean {a[sel, 1].mean():.1f}")
# timing of the product kernel for reference
for _ in range(3): knn_mean_edge(xyz, K=9)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): knn_mean_edge(xyz, K=9)
e1.record(); torch.cuda.synchronize()
print(f"whole k-NN (grids + query, accounting build): {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per call")
