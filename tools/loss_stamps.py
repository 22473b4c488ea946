"""Diagnostic (GPU box): per-wave lifetime of sg_photo_kernel from in-kernel clock stamps.  Needs a library built with -DSG_LOSS_STAMP
(SINGS_HIP_LIB=build/lib_stamp.so python tools/loss_stamps.py [WxH])."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sings_amd import _lib
from sings_amd.photo_loss import PhotoLossEngine
W, H = ([tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(1920, 1080)])[0]
dev = torch.device("cuda:0")
e = PhotoLossEngine(W, H, dev)
raw = torch.rand((3, H, W), device=dev); gt = torch.rand((3, H, W), device=dev); m = torch.ones((H, W), device=dev); bg = torch.zeros(3, device=dev)
for _ in range(20): e(raw, gt, m, bg)
torch.cuda.synchronize()
lib = _lib.load()
n = 8192 * 3 * 4
buf = (C.c_ulonglong * n)()
lib.sg_debug_loss_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sg_debug_loss_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 3, 4).astype(np.int64)
live = s[:, :, 3] > 0
s = s[live.all(1)]
print("workgroups", len(s))
c0, r0, c1, r1 = s[..., 0], s[..., 1], s[..., 2], s[..., 3]
t0 = r0.min()
print("kernel span us (100 MHz clock)", (r1.max() - t0) / 100.0)
print("start offsets us: median %.2f p99 %.2f max %.2f" % tuple(np.percentile((r0 - t0) / 100.0, [50, 99, 100])))
life = (r1 - r0) / 100.0
print("wave life us: min %.1f median %.1f p99 %.1f max %.1f" % tuple(np.percentile(life, [0, 50, 99, 100])))
clk = (c1 - c0) / np.maximum(r1 - r0, 1) * 100.0
print("shader clock MHz: median %.0f min %.0f max %.0f" % (np.median(clk), clk.min(), clk.max()))
for w, nme in enumerate("HVG"):
    print(nme, "life median %.1f us" % np.median(life[:, w]), "end offset median %.1f" % np.median((r1[:, w] - t0) / 100.0))
