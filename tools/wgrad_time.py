"""Times sg_weight_grad for the decoders' layer shapes (GPU box): python tools/wgrad_time.py"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sings_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
tot = 0.0
for Cout, Cin in ((128, 96), (128, 128), (128, 128), (3, 128), (1, 128), (64, 96), (64, 64), (48, 64), (1, 64)):
    dz = torch.randn(N, Cout, device=dev); x = torch.randn(N, Cin, device=dev)
    dW = torch.empty(Cout, Cin, device=dev); db = torch.empty(Cout, device=dev)
    ws = torch.empty(int(lib.sg_weight_grad_ws_bytes(N, Cout, Cin)), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    f = lambda: lib.sg_weight_grad(N, Cout, Cin, p(dz), p(x), p(ws), p(dW), p(db), st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 50 * 1e6
    err = float((dW - dz.t() @ x).abs().max() / (dz.t() @ x).abs().max())
    tot += us
    print(f"dW [{Cout:3d} x {Cin:3d}] over {N} rows: {us:7.1f} us  {2.0 * N * Cout * Cin / us * 1e-6:6.1f} TFLOP/s  rel err {err:.1e}")
print(f"all nine layers: {tot:.0f} us")
