import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sings_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev).cuda_stream
p = lambda t: None if t is None else t.data_ptr()
for N, ci, co, act in [(64, 128, 128, 1), (700, 64, 128, 1), (100, 96, 64, 0)]:
    torch.manual_seed(0)
    x = torch.randn(N, ci, device=dev); W = torch.randn(co, ci, device=dev) * 0.1; b = torch.randn(co, device=dev)
    h = torch.full((N, co), 7.0, device=dev); aux = torch.full((N, co), 9.0, device=dev)
    rc = lib.sg_linear_forward(N, ci, co, act, p(x), p(W), p(b), None, p(aux), p(h), st)
    torch.cuda.synchronize()
    z = x.double() @ W.double().T + b.double()
    ref = torch.nn.functional.gelu(z) if act == 1 else z
    d = (h.double() - ref).abs()
    print(N, ci, co, act, "rc", rc, "max err", d.max().item(), "untouched h", (h == 7.0).sum().item(), "untouched aux", (aux == 9.0).sum().item())
    bad = (d > 1e-4).nonzero()
    print(" bad count", bad.shape[0], bad[:8].tolist())
    print(" h[0,:4]", h[0, :4].tolist(), "ref", ref[0, :4].tolist())
