"""Generates tests/golden/photo_loss_golden.npz by IMPORTING the reference's own loss functions (read-only at
/root/reference) in the build container.  Only the vectors ship.

    python tests/golden/gen_photo_loss_golden.py

sings/rec/losses/utils.py imports pytorch3d at module level for its (unrelated) regularisers; pytorch3d is not installed
here, so empty placeholder modules satisfy that import -- l1_loss / ssim / create_window below are pure torch and run
exactly as shipped.  The composite + weighting of loss.py:55-69 is applied here with the same expressions.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
for name in ("pytorch3d", "pytorch3d.ops"):
    sys.modules.setdefault(name, types.ModuleType(name))
for fn in ("knn_points", "laplacian", "cot_laplacian", "norm_laplacian"):
    setattr(sys.modules["pytorch3d.ops"], fn, None)
sp = types.ModuleType("sings.rec.utils.body_model.smpl_parsing"); sp.parse_weights = None
sys.modules.setdefault("sings.rec.utils.body_model.smpl_parsing", sp)
from sings.rec.losses import utils as ru                         # noqa: E402

out = {}
rs = np.random.RandomState(77)
for tag, (H, W) in (("a", (48, 64)), ("b", (37, 53)), ("c", (96, 80))):
    raw = (rs.uniform(-0.2, 1.2, (3, H, W))).astype(np.float32)              # exercises the clamp on both sides
    gt = rs.uniform(0, 1, (3, H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = (((xx - W / 2) / (W / 3)) ** 2 + ((yy - H / 2) / (H / 2.5)) ** 2 < 1).astype(np.float32)
    mask[rs.uniform(size=mask.shape) < 0.05] = 0.5                            # soft mask values occur at silhouettes
    bg = rs.uniform(0, 1, 3).astype(np.float32)
    l1_w, ssim_w = 0.8, 0.2
    t_raw = torch.from_numpy(raw).requires_grad_(True)
    t_mask = torch.from_numpy(mask).unsqueeze(0)
    pred = torch.clamp(t_raw, 0.0, 1.0)                                       # gs_renderer_single.py:96
    gt_img = torch.from_numpy(gt) * t_mask + torch.from_numpy(bg)[:, None, None] * (1. - t_mask)   # loss.py:58
    Ll1 = ru.l1_loss(pred, gt_img, t_mask)
    sm = ru.ssim(pred, gt_img)
    loss_ssim = (1.0 - sm) * (t_mask.sum() / (pred.shape[-1] * pred.shape[-2]))
    total = l1_w * Ll1 + ssim_w * loss_ssim
    total.backward()
    out.update({f"{tag}_raw": raw, f"{tag}_gt": gt, f"{tag}_mask": mask, f"{tag}_bg": bg,
                f"{tag}_l1": np.float32(Ll1.item()), f"{tag}_ssim_mean": np.float32(sm.item()),
                f"{tag}_loss_l1": np.float32((l1_w * Ll1).item()), f"{tag}_loss_ssim": np.float32((ssim_w * loss_ssim).item()),
                f"{tag}_grad": t_raw.grad.numpy().copy(), f"{tag}_gt_img": gt_img.detach().numpy()})
out["weights"] = np.array([0.8, 0.2], np.float32)
out["window_1d"] = ru.gaussian(11, 1.5).numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "photo_loss_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("a_")})
