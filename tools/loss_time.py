"""Times sg_photo_loss (forward + gradient) on the GPU box: python tools/loss_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from sings_amd.photo_loss import PhotoLossEngine
dev=torch.device("cuda:0")
for (W,H) in ((1920,1080),(512,896)):
    e=PhotoLossEngine(W,H,dev); raw=torch.rand((3,H,W),device=dev); gt=torch.rand((3,H,W),device=dev); m=torch.ones((H,W),device=dev); bg=torch.zeros(3,device=dev)
    for _ in range(5): e(raw,gt,m,bg)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(100): e(raw,gt,m,bg)
    torch.cuda.synchronize(); print(W,H,"loss fwd+grad us", (time.perf_counter()-t0)*1e4)
