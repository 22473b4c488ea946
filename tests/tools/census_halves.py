"""Census for a finer backward granule (GPU box): for a sample of tiles of the cfg3 scene, how many (entry, quadrant) passes does
the quadrant-per-wave backward walk, and how many rounds would a wave need if its two 32-lane halves walked the lists of the two
4x8-pixel halves of the quadrant in lock-step (rounds = max of the two list lengths)?  Plain torch on the GPU; projection as in
oracle/raster_torch.py.  python tests/tools/census_halves.py [tiles]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sings_amd.scene import synthetic_scene

dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
ntiles = int(sys.argv[1]) if len(sys.argv) > 1 else 600
s = synthetic_scene(200000, 1920, 1080, 3, 3)
W, H = 1920, 1080
T = lambda k: torch.as_tensor(np.asarray(s[k]), dtype=torch.float32, device=dev)
p, op, sc, q = T("means3D"), T("opacities").reshape(-1), T("scales"), T("rotations")
Vm, Pm = T("viewmatrix").reshape(4, 4), T("projmatrix").reshape(4, 4)
tanx, tany = float(s["tanfovx"]), float(s["tanfovy"])
fx, fy = W / (2 * tanx), H / (2 * tany)
xf = lambda M, k: M[0, k] * p[:, 0] + M[1, k] * p[:, 1] + M[2, k] * p[:, 2] + M[3, k]
t = torch.stack([xf(Vm, 0), xf(Vm, 1), xf(Vm, 2)], 1)
ph = torch.stack([xf(Pm, 0), xf(Pm, 1), xf(Pm, 2), xf(Pm, 3)], 1)
front = t[:, 2] > 0.2
pw = 1.0 / (ph[:, 3] + 1e-7)
ndc = ph[:, :2] * pw[:, None]
r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                 2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
RS = R * sc[:, None, :]
Sig = RS @ RS.transpose(1, 2)
tz = torch.where(front, t[:, 2], torch.ones_like(t[:, 2]))
tx = torch.clamp(t[:, 0] / tz, -1.3 * tanx, 1.3 * tanx) * tz
ty = torch.clamp(t[:, 1] / tz, -1.3 * tany, 1.3 * tany) * tz
zero = torch.zeros_like(tx)
J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(-1, 2, 3)
M2 = J @ Vm[:3, :3].t()
cov = M2 @ Sig @ M2.transpose(1, 2)
a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
det = a * c - b * b
ok = front & (det != 0)
cx, cy, cz = c / det, -b / det, a / det
mid = 0.5 * (a + c)
rad = torch.ceil(3 * torch.sqrt(mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))))
pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)
gx, gy = (W + 15) // 16, (H + 15) // 16
x0 = ((pix[:, 0] - rad) / 16).trunc().clamp(0, gx); y0 = ((pix[:, 1] - rad) / 16).trunc().clamp(0, gy)
x1 = ((pix[:, 0] + rad + 15) / 16).trunc().clamp(0, gx); y1 = ((pix[:, 1] + rad + 15) / 16).trunc().clamp(0, gy)
ok = ok & ((x1 - x0) * (y1 - y0) > 0)
g = torch.Generator(device="cpu").manual_seed(1)
tiles = torch.randperm(gx * gy, generator=g)[:ntiles].tolist()
py, px = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
quad = ((py // 8) * 2 + (px // 8)).reshape(-1)                 # quadrant of a pixel
half = (((py % 8) // 4)).reshape(-1)                            # upper / lower 4 rows of the quadrant
Q = Hmax = Hsum = E = 0
for tl in tiles:
    ty_, tx_ = tl // gx, tl % gx
    m = ok & (x0 <= tx_) & (x1 > tx_) & (y0 <= ty_) & (y1 > ty_)
    idx = m.nonzero().reshape(-1)
    if idx.numel() == 0:
        continue
    idx = idx[torch.argsort(t[idx, 2])]
    X = (tx_ * 16 + px).reshape(-1).float(); Y = (ty_ * 16 + py).reshape(-1).float()
    dx = pix[idx, 0][:, None] - X[None]; dy = pix[idx, 1][:, None] - Y[None]
    power = -0.5 * (cx[idx][:, None] * dx * dx + cz[idx][:, None] * dy * dy) - cy[idx][:, None] * dx * dy
    alpha = torch.clamp(op[idx][:, None] * torch.exp(power), max=0.99)
    valid = (power <= 0) & (alpha >= 1 / 255)
    al = torch.where(valid, alpha, torch.zeros_like(alpha))
    Tacc = torch.cumprod(1 - al, 0)
    Tbefore = torch.cat([torch.ones_like(Tacc[:1]), Tacc[:-1]], 0)
    contrib = valid & (Tbefore * (1 - al) >= 1e-4)              # blended (not the terminating entry): upstream's rule
    E += idx.numel()
    for qd in range(4):
        cq = contrib[:, quad == qd]
        anyq = cq.any(1)
        top = contrib[:, (quad == qd) & (half == 0)].any(1).sum().item()
        bot = contrib[:, (quad == qd) & (half == 1)].any(1).sum().item()
        Q += anyq.sum().item(); Hmax += max(top, bot); Hsum += top + bot
print(f"{len(tiles)} tiles, {E} list entries: (entry, quadrant) passes with a contribution {Q}; lock-step half-quadrant rounds {Hmax} "
      f"({Hmax / Q:.3f} of the passes); half-quadrant entries {Hsum} ({Hsum / (2 * Q):.3f} per half-slot)")
