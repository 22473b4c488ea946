"""GPU test of the host-side mirror of the reference's render glue:
same keys, dtypes and side outputs as sings/rec/renderer/gs_renderer_single.py:12-107."""
import math

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from sings_amd.camera import make_camera
from sings_amd.scene import synthetic_scene

pytestmark = pytest.mark.gpu


def _data(dev, W, H):
    cam = make_camera(np.eye(4, dtype=np.float32), 1.2 * W, 1.2 * W, W / 2, H / 2, W, H)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return cam, dict(fovx=cam["fovx"], fovy=cam["fovy"], image_height=H, image_width=W,
                     world_view_transform=t(cam["world_view_transform"]), full_proj_transform=t(cam["full_proj_transform"]),
                     camera_center=t(cam["camera_center"]))


def test_get_render_pkg_keys_dtypes_and_densification_outputs():
    from sings_amd.renderer import get_render_pkg, get_render_pkgs
    dev = torch.device("cuda:0")
    W, H = 160, 96
    s = synthetic_scene(2500, W, H, 2, 6)
    cam, data = _data(dev, W, H)
    req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
    gs = dict(xyz=req(s["means3D"]), shs=req(s["shs"]), opacity=req(s["opacities"]), scales=req(s["scales"]),
              rotq=req(s["rotations"]), active_sh_degree=2)
    bg = torch.rand(3, device=dev)                                 # gs_trainer.py:238
    pkg = get_render_pkg(data, gs, bg)
    assert set(pkg) >= {"render", "viewspace_points", "visibility_filter", "radii", "human_visibility_filter", "human_radii"}
    assert pkg["render"].shape == (3, H, W) and pkg["render"].dtype == torch.float32
    assert float(pkg["render"].detach().min()) >= 0.0 and float(pkg["render"].detach().max()) <= 1.0      # clamp(0,1), :96
    assert pkg["radii"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool
    assert torch.equal(pkg["visibility_filter"], pkg["radii"] > 0)
    o = ro.forward(s["means3D"], s["opacities"], cam["world_view_transform"], cam["full_proj_transform"], cam["camera_center"],
                   W, H, math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), bg.cpu().numpy(), scales=s["scales"],
                   rotations=s["rotations"], shs=s["shs"], sh_degree=2)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), o["radii"])
    strict = o["margin"] >= 2e-5
    assert np.abs(pkg["render"].detach().cpu().numpy() - np.clip(o["color"], 0, 1)).max(0)[strict].max() <= 1e-5
    # the clamp zeroes dL/dimage on saturated pixels before the op's backward (SURVEY App. A.5 last bullet)
    sat = (o["color"] <= 0) | (o["color"] >= 1)
    dLn = s["dL_dimage"][:, :H, :W].copy(); dLn[:, ~strict] = 0
    (pkg["render"] * torch.from_numpy(dLn).to(dev)).sum().backward()      # clamp's own autograd zeroes saturated pixels
    dLn[sat] = 0
    g = ro.backward(o, dLn)
    vg = pkg["viewspace_points"].grad                                # consumer: sings_hybrid.py:1013-1015
    assert vg.shape == (2500, 3) and float(vg[:, 2].abs().max()) == 0.0
    ok = np.abs(vg.cpu().numpy() - g["dL_dmean2D"]) <= 5e-4 * np.abs(g["dL_dmean2D"]) + 1e-4 * np.abs(g["dL_dmean2D"]).max()
    assert ok.mean() > 0.9995         # pixels within rounding of the 0 / 1 clamp may saturate on one side only
    # several avatars in one frame: concatenation then one raster call (gs_renderer_multiple.py:12-68)
    half = lambda d, a, b: {k: (v[a:b] if torch.is_tensor(v) else v) for k, v in d.items()}
    with torch.no_grad():
        zero = torch.zeros(3, device=dev)
        p2 = get_render_pkgs(data, [half(gs, 0, 1000), half(gs, 1000, 2500)], [zero, zero], [None, None], bg)
        p1 = get_render_pkg(data, gs, bg)
    assert torch.equal(p1["render"], p2["render"]) and torch.equal(p1["radii"], p2["radii"])


def test_get_render_pkg_fused_matches_unfused():
    from oracle import lbs_oracle as lo
    from sings_amd.renderer import get_render_pkg, get_render_pkg_fused
    from sings_amd.scene import avatar_scene
    from sings_amd.body import joint_transforms
    dev = torch.device("cuda:0")
    s = avatar_scene(N=20000, J=52, W=256, H=448, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cam = make_camera(np.eye(4, dtype=np.float32), 2500.0, 2500.0, 128, 224, 256, 448)
    data = dict(fovx=cam["fovx"], fovy=cam["fovy"], image_height=448, image_width=256,
                world_view_transform=t(cam["world_view_transform"]), full_proj_transform=t(cam["full_proj_transform"]),
                camera_center=t(cam["camera_center"]))
    pose = np.zeros(156, np.float32); pose[3:72] = np.random.RandomState(1).normal(0, 0.25, 69)
    A = joint_transforms(t(pose), t(s["joints_rest"]), tuple(s["parents"]))
    canon = dict(xyz_canon=t(s["xyz_canon"]), rotmat_canon=None, scales=t(s["scales"]), opacity=t(s["opacities"]),
                 shs=t(s["shs"]), lbs_weights=t(s["lbs_weights"]), active_sh_degree=0)
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        fused = get_render_pkg_fused(data, canon, A, bg, smpl_scale=t(s["smpl_scale"]), transl=t(s["transl"]), return_posed=True)
        gs = dict(xyz=fused["xyz"], shs=canon["shs"], opacity=canon["opacity"], scales=fused["scales"], rotq=fused["rotq"],
                  active_sh_degree=0)
        plain = get_render_pkg(data, gs, bg)
    assert torch.equal(fused["render"], plain["render"]) and torch.equal(fused["radii"], plain["radii"])
    assert int(fused["visibility_filter"].sum()) > 15000
    # isotropic canonical rotation = identity -> posed quaternion of the blended joint rotation (non-unit is expected)
    xyz_o, q_o, sc_o, _ = lo.deform_gaussians(torch.from_numpy(s["xyz_canon"]), torch.eye(3)[None].repeat(20000, 1, 1),
                                              torch.from_numpy(s["scales"]), torch.from_numpy(s["lbs_weights"]), A.cpu(),
                                              smpl_scale=torch.from_numpy(s["smpl_scale"]), transl=torch.from_numpy(s["transl"]))
    assert np.abs(fused["xyz"].cpu().numpy() - xyz_o.numpy()).max() < 5e-5
    assert np.abs(fused["rotq"].cpu().numpy() - q_o.numpy()).max() < 5e-5


def test_animate_chunk_matches_per_frame_fused_calls():
    from sings_amd.body import joint_transforms
    from sings_amd.posed import animate_chunk
    from sings_amd.renderer import get_render_pkg_fused
    from sings_amd.scene import avatar_scene
    dev = torch.device("cuda:0")
    s = avatar_scene(N=8000, J=24, W=128, H=224, seed=5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cam = make_camera(np.eye(4, dtype=np.float32), 1250.0, 1250.0, 64, 112, 128, 224)
    data = dict(fovx=cam["fovx"], fovy=cam["fovy"], image_height=224, image_width=128,
                world_view_transform=t(cam["world_view_transform"]), full_proj_transform=t(cam["full_proj_transform"]),
                camera_center=t(cam["camera_center"]))
    rs = np.random.RandomState(2)
    poses = t(rs.normal(0, 0.2, (5, 72)).astype(np.float32))
    transl = t(np.tile(s["transl"], (5, 1)) + rs.normal(0, 0.02, (5, 3)).astype(np.float32))
    jr = t(s["joints_rest"])
    canon = dict(xyz_canon=t(s["xyz_canon"]), rotmat_canon=None, scales=t(s["scales"]), opacity=t(s["opacities"]),
                 shs=t(s["shs"]), lbs_weights=t(s["lbs_weights"]), active_sh_degree=0)
    A_cano = joint_transforms(torch.zeros(72, device=dev), jr, tuple(s["parents"]))          # canonical = rest pose
    bg = torch.ones(3, device=dev)
    imgs = dict(animate_chunk(canon, poses, jr, A_cano, data, bg, transl=transl, parents=tuple(s["parents"]), chunk_size=2))
    assert sorted(imgs) == [0, 1, 2, 3, 4]
    for f in range(5):
        A = joint_transforms(poses[f], jr, tuple(s["parents"])) @ torch.inverse(A_cano)
        with torch.no_grad():
            ref = get_render_pkg_fused(data, canon, A, bg, transl=transl[f])["render"]
        assert torch.allclose(imgs[f], ref, atol=2e-6)
        assert float((imgs[f] < 0.999).float().mean()) > 0.02          # the avatar is in view
    # three frames in flight on three streams (pre-allocated engines): the same images, bit for bit
    imgs3 = dict(animate_chunk(canon, poses, jr, A_cano, data, bg, transl=transl, parents=tuple(s["parents"]), chunk_size=4,
                               streams=3))
    assert sorted(imgs3) == [0, 1, 2, 3, 4]
    for f in range(5):
        assert torch.equal(imgs3[f], imgs[f])
    # K consecutive frames per DISPATCH (SkinnedFramesEngine): batches of 2 + 2 + 1 (padded) on two streams, one padded batch of 8
    for K, streams in ((2, 2), (8, 1), (3, 3)):
        imgsK = dict(animate_chunk(canon, poses, jr, A_cano, data, bg, transl=transl, parents=tuple(s["parents"]), chunk_size=4,
                                   streams=streams, frames_per_launch=K))
        assert sorted(imgsK) == [0, 1, 2, 3, 4]
        for f in range(5):
            assert torch.equal(imgsK[f], imgs[f]), (K, streams, f)
    # every frame its own camera (list of camera dicts): stacked cameras inside a launch
    cams = []
    for f in range(5):
        V = np.eye(4, dtype=np.float32); V[3, 0] = 0.02 * f
        cm = make_camera(V, 1250.0, 1250.0, 64, 112, 128, 224)
        cams.append(dict(fovx=cm["fovx"], fovy=cm["fovy"], image_height=224, image_width=128,
                         world_view_transform=t(cm["world_view_transform"]), full_proj_transform=t(cm["full_proj_transform"]),
                         camera_center=t(cm["camera_center"])))
    ref_c = dict(animate_chunk(canon, poses, jr, A_cano, cams, bg, transl=transl, parents=tuple(s["parents"]), chunk_size=4))
    got_c = dict(animate_chunk(canon, poses, jr, A_cano, cams, bg, transl=transl, parents=tuple(s["parents"]), chunk_size=4,
                               streams=2, frames_per_launch=4))
    for f in range(5):
        assert torch.equal(got_c[f], ref_c[f]), f
    assert not torch.equal(ref_c[0], ref_c[4])


def test_render_then_fused_photometric_loss_matches_torch_chain():
    """render_pkg['render_raw'] -> sg_photo_loss -> rasterizer backward  ==  the reference's chain
    clamp -> l1_loss + ssim (torch ops, here the oracle's restatement on the GPU tensors) -> rasterizer backward."""
    from oracle import photo_loss_oracle as plo
    from sings_amd.photo_loss import photometric_loss
    from sings_amd.renderer import get_render_pkg
    dev = torch.device("cuda:0")
    W, H = 160, 96
    s = synthetic_scene(2500, W, H, 2, 6)
    cam, data = _data(dev, W, H)
    rs = np.random.RandomState(3)
    gt = torch.from_numpy(rs.uniform(0, 1, (3, H, W)).astype(np.float32)).to(dev)
    mask = torch.from_numpy((rs.uniform(size=(H, W)) < 0.7).astype(np.float32)).to(dev)
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    grads = []
    for fused in (True, False):
        req = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)
        gs = dict(xyz=req(s["means3D"]), shs=req(s["shs"]), opacity=req(s["opacities"]), scales=req(s["scales"]),
                  rotq=req(s["rotations"]), active_sh_degree=2)
        pkg = get_render_pkg(data, gs, bg)
        if fused:
            ld, _ = photometric_loss(pkg["render_raw"], gt, mask, bg, 0.8, 0.2)
            loss = ld["l1"] + ld["ssim"]
        else:
            o = plo.photometric_loss(pkg["render_raw"], gt, mask, bg, 0.8, 0.2)
            loss = o["l1"] + o["ssim"]
        loss.backward()
        grads.append([loss.item()] + [gs[k].grad.cpu().numpy() for k in ("xyz", "shs", "opacity", "scales", "rotq")])
    assert abs(grads[0][0] - grads[1][0]) <= 2e-6 * abs(grads[1][0])
    for a, b in zip(grads[0][1:], grads[1][1:]):
        scale = np.abs(b).max()
        assert (np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-5 * scale).all()


def test_render_glue_against_the_references_own_render():
    """tests/golden/render_glue_golden.npz holds what the REFERENCE's get_render_pkg / render (gs_renderer_single.py:12-107)
    returned in the build container (CPU, `diff_gaussian_rasterization` = an oracle-backed stand-in, gen_render_glue_golden.py):
    same keys and dtypes, same image after the clamp, same radii / visibility, same viewspace_points.grad and input gradients
    through the clamp; 2-D feats go to colors_precomp, bg_color=None means black, scaling_modifier is passed on."""
    import os
    from sings_amd.renderer import get_render_pkg, render
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_glue_golden.npz"))
    dev = torch.device("cuda:0")
    N, W, H, deg, seed = (int(v) for v in G["case"])
    s = synthetic_scene(N, W, H, deg, seed)
    cam, data = _data(dev, W, H)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    req = lambda a: t(a).requires_grad_(True)
    gs = dict(xyz=req(s["means3D"]), shs=req(s["shs"]), opacity=req(s["opacities"]), scales=req(s["scales"]), rotq=req(s["rotations"]),
              active_sh_degree=deg)
    pkg = get_render_pkg(data, gs, t(G["bg"]))
    assert set(pkg) >= set(str(k) for k in G["sh_keys"])
    for kd in G["sh_dtypes"]:
        k, d = str(kd).split(":")
        assert str(pkg[k].dtype) == d, (k, pkg[k].dtype, d)
    strict = G["sh_strict"]
    assert np.abs(pkg["render"].detach().cpu().numpy() - G["sh_render"]).max(0)[strict].max() <= 1e-5
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), G["sh_radii"])
    np.testing.assert_array_equal(pkg["visibility_filter"].cpu().numpy(), G["sh_visibility_filter"])
    assert pkg["human_radii"] is pkg["radii"] and pkg["human_visibility_filter"] is pkg["visibility_filter"]
    (pkg["render"] * t(G["sh_dL"])).sum().backward()

    def close(a, b, what):
        a = a.cpu().numpy().astype(np.float64); b = b.astype(np.float64)
        ok = np.abs(a - b) <= 5e-4 * np.abs(b) + 1e-4 * np.abs(b).max()
        assert ok.mean() > 0.9995, (what, np.abs(a - b).max())    # pixels within rounding of the 0 / 1 clamp may saturate on one side only
    close(pkg["viewspace_points"].grad, G["sh_viewspace_grad"], "viewspace")
    for k in ("xyz", "shs", "opacity", "scales", "rotq"):
        close(gs[k].grad, G[f"sh_grad_{k}"], k)
    with torch.no_grad():
        p2 = render(t(s["means3D"]), t(G["rgb_feats"]), t(s["opacities"]), t(s["scales"]), t(s["rotations"]), data, scaling_modifier=0.8)
    assert set(p2) >= set(str(k) for k in G["rgb_keys"])
    np.testing.assert_array_equal(p2["radii"].cpu().numpy(), G["rgb_radii"])
    assert np.abs(p2["render"].cpu().numpy() - G["rgb_render"]).max(0)[G["rgb_strict"]].max() <= 1e-5
    # the multi-avatar twin (gs_renderer_multiple.py:12-68): two avatars, translated in place, one raster call
    from sings_amd.renderer import get_render_pkgs
    cut = int(G["multi_cut"])
    part = lambda a, b: dict(xyz=t(s["means3D"][a:b]) + 0, shs=t(s["shs"][a:b]), opacity=t(s["opacities"][a:b]),
                             scales=t(s["scales"][a:b]), rotq=t(s["rotations"][a:b]), active_sh_degree=deg)
    outs = [part(0, cut), part(cut, N)]
    with torch.no_grad():
        pm = get_render_pkgs(data, outs, [t(G["multi_trans"][0]), t(G["multi_trans"][1])], [None, None], t(G["bg"]))
    assert set(pm) >= set(str(k) for k in G["multi_keys"])
    np.testing.assert_array_equal(pm["radii"].cpu().numpy(), G["multi_radii"])
    assert np.abs(pm["render"].cpu().numpy() - G["multi_render"]).max(0)[G["multi_strict"]].max() <= 1e-5
    assert torch.allclose(outs[1]["xyz"], t(s["means3D"][cut:]) + t(G["multi_trans"][1])[None])     # translated in place (:25-27)
    with pytest.raises(ValueError):
        get_render_pkgs(data, outs, [t(G["multi_trans"][0])] * 2, [None, None], t(G["bg"]), render_mode="single-person")


@pytest.mark.gpu
@pytest.mark.parametrize("with_rot,shared_transl,per_frame_cameras", [(False, False, False), (True, True, True)])
def test_fused_chunk_render_pkg_equals_per_frame_fused_calls(with_rot, shared_transl, per_frame_cameras):
    """get_render_pkgs_fused (K frames, one differentiable call) against K get_render_pkg_fused calls: images, radii, per-frame
    screen-space gradients, dL/dA bit for bit; the canonical-Gaussian gradients (summed over the frames inside the kernel, in frame
    order; autograd adds the per-frame calls' in its own order) to 1e-6 of their scale."""
    from sings_amd.body import joint_transforms
    from sings_amd.renderer import get_render_pkg_fused, get_render_pkgs_fused
    from sings_amd.scene import avatar_scene
    dev = torch.device("cuda:0")
    K, N, J = 5, 9000, 24
    s = avatar_scene(N=N, J=J, W=128, H=224, seed=7, isotropic=not with_rot)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = np.random.RandomState(3)
    cams = []
    for f in range(K):
        V = np.eye(4, dtype=np.float32); V[3, 0] = 0.02 * f if per_frame_cameras else 0.0
        cm = make_camera(V, 1250.0, 1250.0, 64, 112, 128, 224)
        cams.append(dict(fovx=cm["fovx"], fovy=cm["fovy"], image_height=224, image_width=128,
                         world_view_transform=t(cm["world_view_transform"]), full_proj_transform=t(cm["full_proj_transform"]),
                         camera_center=t(cm["camera_center"])))
    jr = t(s["joints_rest"])
    A0 = torch.stack([joint_transforms(t(rs.normal(0, 0.2, J * 3).astype(np.float32)), jr, tuple(s["parents"])) for _ in range(K)])
    tr0 = t(s["transl"]) if shared_transl else t(np.tile(s["transl"], (K, 1)) + rs.normal(0, 0.02, (K, 3)).astype(np.float32))
    w = t(rs.normal(0, 1, (K, 3, 224, 128)).astype(np.float32))
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)

    def leaves():
        c = dict(xyz_canon=t(s["xyz_canon"]).requires_grad_(True),
                 rotmat_canon=t(rs2.normal(size=(N, 6)).astype(np.float32)).requires_grad_(True) if with_rot else None,
                 scales=t(s["scales"]).requires_grad_(True), opacity=t(s["opacities"]).requires_grad_(True),
                 shs=t(s["shs"]).requires_grad_(True), lbs_weights=t(s["lbs_weights"]), active_sh_degree=0)
        return c, A0.clone().requires_grad_(True), tr0.clone().requires_grad_(True)
    rs2 = np.random.RandomState(9)
    c1, A1, tr1 = leaves()
    rs2 = np.random.RandomState(9)
    c2, A2, tr2 = leaves()
    # K single-frame fused calls
    imgs, radii, vsp = [], [], []
    loss = 0
    for f in range(K):
        pkg = get_render_pkg_fused(cams[f] if per_frame_cameras else cams[0], c1, A1[f], bg, transl=tr1 if shared_transl else tr1[f])
        imgs.append(pkg["render_raw"]); radii.append(pkg["radii"]); vsp.append(pkg["viewspace_points"])
        loss = loss + (pkg["render_raw"] * w[f]).sum()
    loss.backward()
    # one K-frame call
    pk = get_render_pkgs_fused(cams if per_frame_cameras else cams[0], c2, A2, bg, transl=tr2)
    (pk["render_raw"] * w).sum().backward()
    for f in range(K):
        assert torch.equal(pk["render_raw"][f], imgs[f]) and torch.equal(pk["radii"][f], radii[f]), f
        assert torch.equal(pk["viewspace_points"].grad[f], vsp[f].grad), f
    assert torch.equal(pk["visibility_filter"], pk["radii"] > 0)
    assert torch.equal(A2.grad, A1.grad)
    close = lambda a, b: float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12
    if shared_transl:
        assert close(tr2.grad, tr1.grad)
    else:
        assert torch.equal(tr2.grad, tr1.grad)
    for key in ("xyz_canon", "scales", "opacity", "shs") + (("rotmat_canon",) if with_rot else ()):
        assert close(c2[key].grad, c1[key].grad), key
        assert float(c1[key].grad.abs().max()) > 0
