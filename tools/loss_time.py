"""Times sg_photo_loss (forward + gradient) on the GPU box: python tools/loss_time.py [WxH ...] [K=n] [avatar]
`avatar`: an avatar-like frame (render = target = background outside an ellipse that covers ~20 % of the image)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from sings_amd.photo_loss import PhotoLossEngine
dev = torch.device("cuda:0")
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(1920, 1080), (512, 896)]
K = ([int(a[2:]) for a in sys.argv[1:] if a.startswith("K=")] or [1])[0]
avatar = "avatar" in sys.argv[1:]
for (W, H) in sizes:
    e = PhotoLossEngine(W, H, dev, K=K)
    sh = (3, H, W) if K == 1 else (K, 3, H, W)
    raw = torch.rand(sh, device=dev); gt = torch.rand(sh, device=dev); bg = torch.tensor([0.25, 0.5, 0.75], device=dev)
    m = torch.ones((H, W) if K == 1 else (K, H, W), device=dev)
    if avatar:
        yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        body = (((xx - W * 0.5) / (W * 0.2)) ** 2 + ((yy - H * 0.5) / (H * 0.32)) ** 2 < 1.0).float()
        m = body if K == 1 else body.expand(K, H, W).contiguous()
        raw = raw * body + bg[:, None, None] * (1 - body)
    for _ in range(5): e(raw, gt, m, bg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): e(raw, gt, m, bg)
    torch.cuda.synchronize(); print(W, H, "K", K, "avatar" if avatar else "noise", "loss fwd+grad us per frame", (time.perf_counter() - t0) * 1e4 / K)
