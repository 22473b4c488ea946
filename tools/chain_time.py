"""Times the SMPL kinematic chain (GPU box): python tools/chain_time.py -- sg_joint_transforms vs the torch chain."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd import body
dev = torch.device("cuda:0")
for J, B in ((24, 1), (52, 1), (52, 16), (52, 128)):
    parents = body.SMPL_PARENTS if J == 24 else (-1,) + tuple((i - 1) // 2 for i in range(1, J))
    pose = (0.3 * torch.randn(B, J * 3, device=dev)).requires_grad_(True); jr = torch.randn(J, 3, device=dev)
    g = torch.randn(B, J, 4, 4, device=dev)

    def hip(bwd):
        A = body.joint_transforms_hip(pose, jr, parents)
        if bwd:
            pose.grad = None; (A * g).sum().backward()

    def eager(bwd):
        A = torch.stack([body._joint_transforms_torch(pose[b], jr, parents) for b in range(B)]) if B == 1 else None
        if A is None:
            from sings_amd import posed
            import sings_amd.posed as P
            A = P.joint_transforms_batch(pose.double(), jr.double(), parents).float()      # (fp64 input -> the torch path)
        if bwd:
            pose.grad = None; (A * g).sum().backward()

    def timeit(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print(f"J={J} B={B}: forward HIP {timeit(lambda: hip(False)):7.1f} us | torch {timeit(lambda: eager(False)):8.1f} us     "
          f"forward+backward HIP {timeit(lambda: hip(True)):7.1f} us | torch {timeit(lambda: eager(True)):8.1f} us")
