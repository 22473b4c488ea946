import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sings_amd.photo_loss import photometric_loss, PhotoLossEngine
dev = torch.device("cuda:0")
W, H = 1920, 1080
raw = torch.rand((3, H, W), device=dev).requires_grad_(True); gt = torch.rand((3, H, W), device=dev); m = torch.ones((H, W), device=dev); bg = torch.zeros(3, device=dev)
def step():
    raw.grad = None
    ld, _ = photometric_loss(raw, gt, m, bg)
    (ld["l1"] + ld["ssim"]).backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print("autograd fwd + bwd us", (time.perf_counter() - t0) * 1e4)
def fwd():
    with torch.no_grad():
        photometric_loss(raw, gt, m, bg)
for _ in range(5): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): fwd()
torch.cuda.synchronize(); print("forward only us", (time.perf_counter() - t0) * 1e4)
