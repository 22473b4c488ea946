"""Per-shape timing of the decoder linear kernels (sg_linear_forward / sg_linear_backward / sg_weight_grad) on one GPU:
the layer shapes of GeometryDecoder(96) / AppearanceDecoder(96) at N points, HIP events around `--iters` back-to-back launches.
Prints us per launch and the algorithmic HBM rate (bytes every launch must move / time).  Usage: python tools/linear_bench.py"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    import torch
    from sings_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    N = a.points
    st = torch.cuda.current_stream(dev).cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    shapes = [(128, 128, 0), (96, 128, 1), (128, 128, 1), (128, 3, 0), (128, 6, 0), (128, 1, 0), (96, 64, 1), (64, 64, 1), (64, 1, 2), (64, 48, 0)]

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3

    print(f"N = {N}")
    for ci, co, act in shapes:
        x = torch.randn(N, ci, device=dev); W = torch.randn(co, ci, device=dev) * 0.1; b = torch.randn(co, device=dev)
        h = torch.empty(N, co, device=dev); aux = torch.empty(N, co, device=dev) if act == 1 else None
        dh = torch.randn(N, co, device=dev); dz = torch.empty(N, co, device=dev); dx = torch.empty(N, ci, device=dev)
        dW = torch.empty(co, ci, device=dev); db = torch.empty(co, device=dev)
        ws = torch.empty(int(lib.sg_weight_grad_ws_bytes(N, co, ci)), dtype=torch.uint8, device=dev)
        tf = timed(lambda: lib.sg_linear_forward(N, ci, co, act, p(x), p(W), p(b), None, p(aux), p(h), st))
        bf = 4 * N * (ci + co * (2 if act == 1 else 1))
        auxb = aux if act == 1 else (h if act == 2 else None)
        tb = timed(lambda: lib.sg_linear_backward(N, ci, co, act, p(auxb), None, p(dh), p(W), p(dz) if act else None, p(dx), st))
        bb = 4 * N * (ci + co * (3 if act else 1))
        tw = timed(lambda: lib.sg_weight_grad(N, co, ci, p(dz), p(x), p(ws), p(dW), p(db), st))
        if os.environ.get("LB_ONLY_WGRAD"):
            print(f"{ci:4d} -> {co:4d}: wgrad {tw:7.1f} us"); continue
        bw = 4 * N * (ci + co)
        print(f"{ci:4d} -> {co:4d} act {act}:  fwd {tf:7.1f} us ({bf / tf / 1e6:5.2f} TB/s)   bwd {tb:7.1f} us ({bb / tb / 1e6:5.2f} TB/s)"
              f"   wgrad {tw:7.1f} us ({bw / tw / 1e6:5.2f} TB/s)")


if __name__ == "__main__":
    main()
