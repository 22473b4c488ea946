"""CPU: the photometric-loss oracle (oracle/photo_loss_oracle.py) against golden vectors produced by the reference's own
l1_loss / ssim (tests/golden/gen_photo_loss_golden.py)."""
import os

import numpy as np
import torch

from oracle import photo_loss_oracle as plo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "photo_loss_golden.npz"))


def test_window_matches_reference():
    w = plo.window(3)
    g = G["window_1d"]                                        # the reference's gaussian(11, 1.5), fp32
    np.testing.assert_array_equal(w[0, 0].numpy(), (g[:, None] * g[None, :]).astype(np.float32))
    assert w.shape == (3, 1, 11, 11) and abs(float(g.sum()) - 1.0) < 1e-6


def test_oracle_matches_reference_losses_and_gradient():
    for tag in "abc":
        raw = torch.from_numpy(G[f"{tag}_raw"]).requires_grad_(True)
        o = plo.photometric_loss(raw, torch.from_numpy(G[f"{tag}_gt"]), torch.from_numpy(G[f"{tag}_mask"]),
                                 torch.from_numpy(G[f"{tag}_bg"]), float(G["weights"][0]), float(G["weights"][1]))
        (o["l1"] + o["ssim"]).backward()
        np.testing.assert_array_equal(o["gt_img"].detach().numpy(), G[f"{tag}_gt_img"])
        assert abs(o["l1_raw"].item() - G[f"{tag}_l1"]) <= 1e-6 * abs(G[f"{tag}_l1"])
        assert abs(o["ssim_mean"].item() - G[f"{tag}_ssim_mean"]) <= 1e-6
        assert abs(o["l1"].item() - G[f"{tag}_loss_l1"]) <= 1e-6 and abs(o["ssim"].item() - G[f"{tag}_loss_ssim"]) <= 1e-6
        np.testing.assert_allclose(raw.grad.numpy(), G[f"{tag}_grad"], rtol=1e-5, atol=1e-9)
