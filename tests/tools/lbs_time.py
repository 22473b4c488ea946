"""Times the stand-alone lbs_extra (GPU box): python tests/tools/lbs_time.py -- HIP op vs the reference formulation in torch eager."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lbs_oracle as lo
from sings_amd.lbs import lbs_extra
dev = torch.device("cuda:0")
for N, J in ((150000, 52), (150000, 24), (500000, 52)):
    rs = np.random.RandomState(0)
    A = torch.from_numpy((np.eye(4)[None, None] + 0.2 * rs.randn(1, J, 4, 4)).astype(np.float32)).to(dev).requires_grad_(True)
    w = torch.rand(N, J, device=dev) ** 4; w = w / w.sum(1, keepdim=True)
    v = torch.randn(1, N, 3, device=dev, requires_grad=True)
    gv, gT = torch.randn(1, N, 3, device=dev), torch.randn(1, N, 4, 4, device=dev)

    def ours(bwd):
        verts, _, T, _, _ = lbs_extra(A, v, None, w, None, disable_posedirs=True)
        if bwd:
            A.grad = None; v.grad = None
            ((verts * gv).sum() + (T * gT).sum()).backward()

    def eager(bwd):
        verts, T = lo.lbs_extra(A, v, w)
        if bwd:
            A.grad = None; v.grad = None
            ((verts * gv).sum() + (T * gT).sum()).backward()

    def timeit(f, n=30):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print(f"N={N} J={J}: forward ours {timeit(lambda: ours(False)):7.1f} us | eager {timeit(lambda: eager(False)):7.1f} us    "
          f"forward+backward ours {timeit(lambda: ours(True)):7.1f} us | eager {timeit(lambda: eager(True)):7.1f} us")
