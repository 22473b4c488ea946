"""GPU parity of sg_photo_loss (through the C ABI) with the reference-generated golden vectors and the CPU oracle.

Tolerances (fp32): the kernel applies the 11x11 window as two separable 11-tap passes with fma, the reference as one
121-tap conv2d -- loss scalars within 2e-6 relative, gradient within 1e-5 of the gradient's max (and 2e-4 relative)."""
import os

import numpy as np
import pytest
import torch

from oracle import photo_loss_oracle as plo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "photo_loss_golden.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close_grad(a, b):
    scale = np.abs(b).max()
    err = np.abs(a - b)
    assert (err <= 2e-4 * np.abs(b) + 1e-5 * scale).all(), (err.max(), scale)


def test_golden_vectors_from_reference():
    from sings_amd.photo_loss import photometric_loss
    dev = _dev()
    for tag in "abc":
        t = lambda k: torch.from_numpy(G[f"{tag}_{k}"]).to(dev)
        raw = t("raw").requires_grad_(True)
        ld, ex = photometric_loss(raw, t("gt"), t("mask"), t("bg"), float(G["weights"][0]), float(G["weights"][1]),
                                  return_images=True)
        (ld["l1"] + ld["ssim"]).backward()
        np.testing.assert_array_equal(ex["gt_img"].cpu().numpy(), G[f"{tag}_gt_img"])
        np.testing.assert_array_equal(ex["pred_img"].cpu().numpy(), np.clip(G[f"{tag}_raw"], 0, 1))
        assert abs(ex["l1_raw"].item() - G[f"{tag}_l1"]) <= 2e-6 * abs(G[f"{tag}_l1"])
        assert abs(ex["ssim_mean"].item() - G[f"{tag}_ssim_mean"]) <= 2e-6
        assert abs(ld["l1"].item() - G[f"{tag}_loss_l1"]) <= 2e-6
        assert abs(ld["ssim"].item() - G[f"{tag}_loss_ssim"]) <= 2e-6
        _close_grad(raw.grad.cpu().numpy(), G[f"{tag}_grad"])


@pytest.mark.parametrize("W,H,seed", [(512, 896, 1), (250, 131, 2), (33, 31, 3), (8, 5, 4)])
def test_vs_oracle_ragged_sizes_and_separate_upstreams(W, H, seed):
    from sings_amd.photo_loss import photometric_loss
    dev = _dev()
    rs = np.random.RandomState(seed)
    raw = rs.uniform(-0.3, 1.3, (3, H, W)).astype(np.float32)
    gt = rs.uniform(0, 1, (3, H, W)).astype(np.float32)
    mask = (rs.uniform(size=(H, W)) < 0.6).astype(np.float32)
    bg = rs.uniform(0, 1, 3).astype(np.float32)
    a, b = 1.7, -0.4                                           # different upstream gradients for the two terms
    r_cpu = torch.from_numpy(raw).requires_grad_(True)
    o = plo.photometric_loss(r_cpu, torch.from_numpy(gt), torch.from_numpy(mask), torch.from_numpy(bg), 0.8, 0.2)
    (a * o["l1"] + b * o["ssim"]).backward()
    r_gpu = torch.from_numpy(raw).to(dev).requires_grad_(True)
    ld, ex = photometric_loss(r_gpu, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)[None],
                              torch.from_numpy(bg).to(dev), 0.8, 0.2)
    (a * ld["l1"] + b * ld["ssim"]).backward()
    assert abs(ld["l1"].item() - o["l1"].item()) <= 2e-6 * abs(o["l1"].item()) + 1e-7
    assert abs(ld["ssim"].item() - o["ssim"].item()) <= 2e-6
    _close_grad(r_gpu.grad.cpu().numpy(), r_cpu.grad.numpy())


def test_full_hd_deterministic_and_flat_image_properties():
    """1920x1080: two runs bitwise identical (no atomics); pred == gt gives ssim 1, l1 0 and a zero gradient."""
    from sings_amd.photo_loss import PhotoLossEngine
    dev = _dev()
    W, H = 1920, 1080
    g = torch.Generator(device="cpu").manual_seed(5)
    raw = torch.rand((3, H, W), generator=g).to(dev) * 1.2 - 0.1
    gt = torch.rand((3, H, W), generator=g).to(dev)
    mask = (torch.rand((H, W), generator=g) < 0.7).float().to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    eng = PhotoLossEngine(W, H, dev)
    g1 = eng(raw, gt, mask, bg).clone(); l1 = eng.losses.clone()
    g2 = eng(raw, gt, mask, bg).clone(); l2 = eng.losses.clone()
    assert torch.equal(g1, g2) and torch.equal(l1, l2)
    ones = torch.ones((H, W), device=dev)
    same = gt.clone()
    ga = eng(same, gt, ones, bg)
    torch.cuda.synchronize()
    assert abs(eng.losses[3].item() - 1.0) <= 1e-6 and eng.losses[2].item() == 0.0
    assert ga.abs().max().item() <= 1e-9


def test_flat_tiles_take_the_short_path_with_the_same_bits():
    """An avatar frame: the body covers part of the image, the rest is background on both sides (render = bg, mask = 0).  Workgroups
    whose whole 42 x 42 halo tile holds ONE value evaluate one 11-tap chain per quantity instead of the tile's (sg_tile_is_flat).
    (i) the result agrees with the oracle like any other image; (ii) a pixel's gradient depends on its 21 x 21 neighbourhood only
    (two 11-tap windows), so every pixel with a flat neighbourhood -- whether its tile went the short way (far from the body) or
    the long way (tiles that also hold part of the body, tiles at the image border) -- must carry the SAME gradient bits."""
    from sings_amd.photo_loss import photometric_loss
    dev = _dev()
    W, H = 352, 288
    rs = np.random.RandomState(11)
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    body = ((xx - 201) ** 2 / 60.0 ** 2 + (yy - 150) ** 2 / 95.0 ** 2) < 1.0
    mask = body.astype(np.float32)
    gt = rs.uniform(0, 1, (3, H, W)).astype(np.float32)                       # (noise outside the mask too: it is multiplied by 0)
    raw = np.where(body[None], rs.uniform(-0.2, 1.2, (3, H, W)), bg[:, None, None]).astype(np.float32)
    r_cpu = torch.from_numpy(raw).requires_grad_(True)
    o = plo.photometric_loss(r_cpu, torch.from_numpy(gt), torch.from_numpy(mask), torch.from_numpy(bg), 0.8, 0.2)
    (o["l1"] + o["ssim"]).backward()
    r_gpu = torch.from_numpy(raw).to(dev).requires_grad_(True)
    ld, _ = photometric_loss(r_gpu, torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev)[None], torch.from_numpy(bg).to(dev), 0.8, 0.2)
    (ld["l1"] + ld["ssim"]).backward()
    assert abs(ld["l1"].item() - o["l1"].item()) <= 2e-6 * abs(o["l1"].item()) + 1e-7
    assert abs(ld["ssim"].item() - o["ssim"].item()) <= 2e-6
    g = r_gpu.grad.cpu().numpy()
    _close_grad(g, r_cpu.grad.numpy())
    # pixels at least 10 px from the body and from the image border: flat 21 x 21 neighbourhood
    from scipy.ndimage import binary_dilation
    near = binary_dilation(body, structure=np.ones((21, 21), bool))
    inner = np.zeros((H, W), bool); inner[10:H - 10, 10:W - 10] = True
    sel = inner & ~near
    ty, tx = yy // 32, xx // 32
    tile_has_body = np.zeros((H // 32 + 1, W // 32 + 1), bool)
    for t_y in range(H // 32 + (H % 32 > 0)):
        for t_x in range(W // 32 + (W % 32 > 0)):
            y0, y1, x0, x1 = max(t_y * 32 - 5, 0), min(t_y * 32 + 37, H), max(t_x * 32 - 5, 0), min(t_x * 32 + 37, W)
            edge = t_y * 32 - 5 < 0 or t_x * 32 - 5 < 0 or t_y * 32 + 37 > H or t_x * 32 + 37 > W
            tile_has_body[t_y, t_x] = body[y0:y1, x0:x1].any() or edge
    long_way = sel & tile_has_body[ty, tx]
    short_way = sel & ~tile_has_body[ty, tx]
    assert long_way.sum() > 2000 and short_way.sum() > 10000, (long_way.sum(), short_way.sum())
    for ch in range(3):
        bits = g[ch].view(np.uint32)
        assert np.unique(bits[sel]).size == 1, (ch, np.unique(bits[sel]).size)


def test_k_frame_autograd_loss_equals_k_single_losses():
    """photometric_loss_frames ([K,3,H,W], one backward launch with a pair of upstream weights PER FRAME) against K
    photometric_loss calls: values and gradients bit for bit, with different weights on every frame's two terms."""
    from sings_amd.photo_loss import photometric_loss, photometric_loss_frames
    dev = _dev()
    K, W, H = 4, 200, 136
    g = torch.Generator(device="cpu").manual_seed(8)
    raw = (torch.rand(K, 3, H, W, generator=g) * 1.4 - 0.2).to(dev)
    gt = torch.rand(K, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(K, H, W, generator=g) > 0.3).float().to(dev)
    bg = torch.tensor([1.0, 0.5, 0.25], device=dev)
    a = torch.tensor([1.0, 0.7, -0.3, 2.0], device=dev); b = torch.tensor([1.0, -1.1, 0.4, 0.0], device=dev)
    r1 = raw.clone().requires_grad_(True)
    tot = 0
    singles = []
    for f in range(K):
        ld, ex = photometric_loss(r1[f], gt[f], mask[f], bg, 0.8, 0.2)
        singles.append((ld["l1"].detach().clone(), ld["ssim"].detach().clone(), ex["l1_raw"].clone(), ex["ssim_mean"].clone()))
        tot = tot + a[f] * ld["l1"] + b[f] * ld["ssim"]
    tot.backward()
    r2 = raw.clone().requires_grad_(True)
    ld, ex = photometric_loss_frames(r2, gt, mask, bg, 0.8, 0.2)
    ((a * ld["l1"]).sum() + (b * ld["ssim"]).sum()).backward()
    for f in range(K):
        assert torch.equal(ld["l1"][f], singles[f][0]) and torch.equal(ld["ssim"][f], singles[f][1])
        assert torch.equal(ex["l1_raw"][f], singles[f][2]) and torch.equal(ex["ssim_mean"][f], singles[f][3])
        assert torch.equal(r2.grad[f], r1.grad[f]), f
    # one target / one mask for all frames; only the L1 term used
    r3 = raw.clone().requires_grad_(True)
    ld3, _ = photometric_loss_frames(r3, gt[0], mask[0], bg, 0.8, 0.2)
    ld3["l1"].sum().backward()
    r4 = raw.clone().requires_grad_(True)
    t4 = 0
    for f in range(K):
        t4 = t4 + photometric_loss(r4[f], gt[0], mask[0], bg, 0.8, 0.2)[0]["l1"]
    t4.backward()
    assert torch.equal(r3.grad, r4.grad)
