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
raw_s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 3, 4).copy()
hwid = (raw_s[..., 1] >> np.uint64(48)).astype(np.int64); xcc = ((raw_s[..., 1] >> np.uint64(44)) & np.uint64(0xf)).astype(np.int64)
raw_s[..., 1] &= np.uint64((1 << 44) - 1)
s = raw_s.astype(np.int64)
live = s[:, :, 3] > 0
sel = live.all(1); s = s[sel]; hwid = hwid[sel]; xcc = xcc[sel]
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

# placement: workgroups per CU (XCC, SE, CU) and lifetime against the load of the CU
cu = xcc[:, 0] * 1000 + ((hwid[:, 0] >> 13) & 7) * 100 + ((hwid[:, 0] >> 8) & 15)
ids, cnt = np.unique(cu, return_counts=True)
print("CUs used", len(ids), "workgroups per CU histogram", dict(zip(*np.unique(cnt, return_counts=True))))
wl = life[:, 0]
for n in np.unique(cnt):
    m = np.isin(cu, ids[cnt == n])
    print("  CUs with %d workgroups: wave life median %.1f max %.1f us" % (n, np.median(wl[m]), wl[m].max()))
simd = (hwid >> 4) & 3
print("waves per SIMD of the busiest CU:", np.bincount(simd[cu == ids[np.argmax(cnt)]].ravel(), minlength=4))
print("life by XCC:", " ".join("%d:%.1f" % (x, np.median(wl[xcc[:, 0] == x])) for x in np.unique(xcc[:, 0])))
se = (hwid[:, 0] >> 13) & 7
print("life by SE :", " ".join("%d:%.1f" % (x, np.median(wl[se == x])) for x in np.unique(se)))
percu = np.array([np.median(wl[cu == i]) for i in ids])
print("per-CU median life: min %.1f median %.1f max %.1f ; within-CU spread (max-min) median %.1f" % (percu.min(), np.median(percu), percu.max(), np.median([wl[cu == i].max() - wl[cu == i].min() for i in ids])))
order = np.argsort(wl)
blk = np.nonzero(sel)[0]
print("slowest 12 workgroups: block ids", blk[order[-12:]], "life", np.round(wl[order[-12:]], 1))
print("fastest 12 workgroups: block ids", blk[order[:12]], "life", np.round(wl[order[:12]], 1))
clkw = clk[:, 0]
print("clock by XCC:", " ".join("%d:%.0f" % (x, np.median(clkw[xcc[:, 0] == x])) for x in np.unique(xcc[:, 0])))
print("corr(life, clock) %.2f" % np.corrcoef(wl, clkw)[0, 1])
