"""K frames of the same Gaussians per call (the ``*_frames`` entry points, include/sings_hip.h; reference shape: SinGS.forward_chunk,
sings/rec/models/sings_hybrid.py:474-569, 16 frames per call, rendered one by one at gs_trainer.py:684-714).

The bar is BIT-IDENTITY with K single-frame calls: every per-frame output (image, radii, screen-space gradient, dL/dA, dL/dtransl)
equals the single-frame call's, and the summed canonical-Gaussian gradient equals what K single-frame calls leave in ONE gradient
buffer when frame 0 writes it and frames 1.. add to it in frame order (``accumulate=1``).  The single-frame path is the one the
oracle tests pin (tests/test_gpu_skinned.py, test_gpu_raster.py), so parity with the oracle carries over to the K-frame calls.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _avatar(N, J, W, H, seed, K, per_frame_camera, with_rot):
    from sings_amd.body import joint_transforms
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import avatar_scene
    dev = _dev()
    s = avatar_scene(N=N, J=J, W=W, H=H, seed=seed, isotropic=not with_rot)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = np.random.RandomState(seed)
    jr = t(s["joints_rest"])
    A = torch.stack([joint_transforms(t(rs.normal(0, 0.2, J * 3).astype(np.float32)), jr, tuple(s["parents"])).reshape(J, 16)
                     for _ in range(K)]).contiguous()
    transl = t((s["transl"][None] + rs.normal(0, 0.03, (K, 3))).astype(np.float32))
    cam = s["cam"]
    view = np.repeat(cam["world_view_transform"][None], K, 0).copy()
    if per_frame_camera:
        view[:, 3, 0] += 0.02 * np.arange(K)                       # every frame its own camera (shifted along x)
    P_T = np.linalg.inv(cam["world_view_transform"]) @ cam["full_proj_transform"]
    proj = np.stack([(v @ P_T).astype(np.float32) for v in view])
    campos = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in view])

    def settings(vm, pm, cp):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5), tanfovy=math.tan(cam["fovy"] * 0.5), bg=t(s["bg"]),
            scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=0, campos=cp, prefiltered=False, debug=False)
    one = [settings(t(view[f]), t(proj[f]), t(campos[f])) for f in range(K)]
    stacked = settings(t(view), t(proj), t(campos)) if per_frame_camera else one[0]
    rot = t(rs.normal(size=(N, 6)).astype(np.float32)) if with_rot else None
    ins = dict(xyz=t(s["xyz_canon"]), w=t(s["lbs_weights"]), sc=t(s["scales"]), op=t(s["opacities"]), sh=t(s["shs"]), rot=rot,
               smpl_scale=t(s["smpl_scale"]))
    dL = t(rs.normal(0, 1, (K, 3, H, W)).astype(np.float32))
    return s, ins, A, transl, one, stacked, dL


@pytest.mark.parametrize("K,per_frame_camera,with_rot", [(1, False, False), (3, False, False), (8, False, False), (4, True, True),
                                                         (16, False, True)])
def test_skinned_frames_equal_single_frame_calls_bit_for_bit(K, per_frame_camera, with_rot):
    from sings_amd.engine import SkinnedEngine, SkinnedFramesEngine
    dev = _dev()
    N, J, W, H = 20000, 52, 160, 288
    s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 3, K, per_frame_camera, with_rot)
    cap = 24 * N
    # ---- reference: K single-frame calls into ONE gradient buffer, frame 0 writes, the others add, in frame order
    ref = SkinnedEngine(N, J, W, H, 16, dev, cap, with_rot=with_rot, rot_width=6)
    ref.throughput = True                                           # (the K-frame calls never split long tiles: same mask layout)
    per_frame = []
    for f in range(K):
        ref.set_camera(one[f])
        ref.set_frame(ins["xyz"], ins["rot"], ins["w"], A[f], ins["smpl_scale"], transl[f])
        R = ref.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
        assert 0 < R <= cap
        ref._chain = (lambda f=f: (f > 0, None, None))
        ref.backward(ins["sh"], ins["op"], ins["sc"], dL[f])
        torch.cuda.synchronize()
        per_frame.append([x.clone() for x in (ref.color, ref.radii, ref.d_means2D, ref.d_A, ref.d_transl)] + [R])
    ref_grad = ref.grad_flat.clone()
    # ---- K frames per call
    eng = SkinnedFramesEngine(N, J, W, H, 16, K, dev, cap, with_rot=with_rot, rot_width=6)
    eng.set_camera(stacked)
    eng.set_frames(ins["xyz"], ins["rot"], ins["w"], A, ins["smpl_scale"], transl)
    for rep in range(2):                                            # twice: the workspaces are reused (SG_FLAG_WS_CLEAN)
        Rs = eng.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
        eng.backward(ins["sh"], ins["op"], ins["sc"], dL)
        torch.cuda.synchronize()
        assert Rs == [p[5] for p in per_frame] and eng.num_rendered() == Rs
        for f in range(K):
            c, r, m2, dA, dt, _ = per_frame[f]
            assert torch.equal(eng.color[f], c), f"image of frame {f}"
            assert torch.equal(eng.radii[f], r) and torch.equal(eng.d_means2D[f], m2), f"radii / means2D of frame {f}"
            assert torch.equal(eng.d_A[f], dA) and torch.equal(eng.d_transl[f], dt), f"dL/dA, dL/dtransl of frame {f}"
        assert torch.equal(eng.grad_flat, ref_grad), "summed canonical-Gaussian gradient"
    assert float(ref_grad.abs().max()) > 0
    if K > 1:                                                       # (the frames really differ: poses, translations)
        assert float((per_frame[0][0] - per_frame[-1][0]).abs().max()) > 1e-3


@pytest.mark.parametrize("K", [1, 4])
def test_direct_binning_of_few_tile_frames_leaves_the_bits_of_the_scatter_path(K):
    """SG_FLAG_LONG_ROWS (``set_camera(long_rows=True)``): on images of few tiles the preprocess writes every pair's key into its
    tile's row (16384 keys) once the workgroup histogram has given the rank; no pair scatter pass, lists of more than 1024 entries
    sorted by the long-list kernels from their rows.  Same images, lists and gradients as the plain path, bit for bit -- on a scene
    with long lists (small splats piled on the image centre), twice (workspaces reused)."""
    from sings_amd.engine import SkinnedEngine, SkinnedFramesEngine
    dev = _dev()
    N, J, W, H = 100000, 52, 160, 288
    s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 5, K, False, True)
    cap = 32 * N
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def run(long_rows):
        if K == 1:
            e = SkinnedEngine(N, J, W, H, 16, dev, cap, with_rot=True, rot_width=6)
            e.set_camera(one[0], long_rows=long_rows)
            e.set_frame(ins["xyz"], ins["rot"], ins["w"], A[0], ins["smpl_scale"], transl[0])
        else:
            e = SkinnedFramesEngine(N, J, W, H, 16, K, dev, cap, with_rot=True, rot_width=6)
            e.set_camera(stacked, long_rows=long_rows)
            e.set_frames(ins["xyz"], ins["rot"], ins["w"], A, ins["smpl_scale"], transl)
        outs = []
        for rep in range(2):
            R = e.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
            e.backward(ins["sh"], ins["op"], ins["sc"], dL[0] if K == 1 else dL)
            torch.cuda.synchronize()
            L = e.L
            b0 = e.binning if K == 1 else e.binning.view(K, L.bin_bytes)[0]
            rg = b0[L.bin_ranges:L.bin_ranges + 8 * T].view(torch.int32).view(-1, 2).clone()
            pl = b0[L.bin_point_list:L.bin_point_list + 4 * int(rg[:, 1].max())].view(torch.int32).clone()
            outs.append((R, e.color.clone(), e.grad_flat.clone(), e.d_A.clone(), e.d_transl.clone(), rg, pl))
        return outs
    plain, direct = run(False), run(True)
    assert int((plain[0][5][:, 1] - plain[0][5][:, 0]).max()) > 1024, "the scene should hold a long list"
    for a, b in zip(plain, direct):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
    assert float(plain[0][2].abs().max()) > 0
    if K == 1:
        # a list that outgrows its row (every splat on the image centre): refused on the device, as the hint promises
        from sings_amd import _lib
        e = SkinnedEngine(N, J, W, H, 16, dev, cap)
        tiny = ins["xyz"] * 0.02
        e.set_camera(one[0]); e.set_frame(tiny, None, ins["w"], A[0], ins["smpl_scale"], transl[0])
        R = e.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
        rg = e.binning[e.L.bin_ranges:e.L.bin_ranges + 8 * T].view(torch.int32).view(-1, 2)
        assert R > 0 and int((rg[:, 1] - rg[:, 0]).max()) > 16384
        c_plain = e.color.clone()
        e.set_camera(one[0], long_rows=True)
        assert e.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True) == _lib.NUM_RENDERED_LONG_LIST
        assert torch.equal(e.color, torch.from_numpy(s["bg"]).to(dev)[:, None, None].expand_as(e.color))
        e.set_camera(one[0])
        assert e.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True) == R and torch.equal(e.color, c_plain)
        # ... and the frame animator, which passes the hint optimistically, falls back to the plain path for good
        from sings_amd.posed import FrameAnimator
        canon = dict(xyz_canon=tiny, rotmat_canon=None, scales=ins["sc"], opacity=ins["op"], shs=ins["sh"], lbs_weights=ins["w"],
                     active_sh_degree=0)
        fa = FrameAnimator(canon, streams=2)
        cam = s["cam"]
        camd = {k_: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if isinstance(v, np.ndarray) else v) for k_, v in cam.items()}
        img = fa.render_round([(camd, A[0].reshape(J, 4, 4), transl[0], ins["smpl_scale"], None)], torch.from_numpy(s["bg"]).to(dev))[0]
        assert fa.long_rows is False and torch.equal(img, torch.clamp(c_plain, 0.0, 1.0))


def test_a_frame_that_overflows_its_workspace_is_background_with_zero_gradient_and_harms_no_other_frame():
    """The pair capacity lies between the smallest and the largest R of the batch: frames that fit render and differentiate as
    ever, frames that do not render the background, report their (too large) R, and contribute ZERO to the summed gradient --
    poisoned workspaces prove that nothing stale is followed (the single-frame contract,
    test_capacity_overflow_is_reported_and_harmless, frame by frame inside one K-frame call)."""
    from sings_amd.engine import SkinnedFramesEngine
    dev = _dev()
    N, J, W, H, K = 20000, 52, 160, 288, 6
    s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 9, K, False, False)
    # frames of different weight: the body comes closer to the camera frame by frame
    transl = transl.clone(); transl[:, 2] -= torch.linspace(0.0, 0.6 * float(transl[0, 2].abs()), K, device=dev) * torch.sign(transl[:, 2])
    big = SkinnedFramesEngine(N, J, W, H, 16, K, dev, 64 * N)
    big.set_camera(one[0]); big.set_frames(ins["xyz"], None, ins["w"], A, ins["smpl_scale"], transl)
    R = big.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
    assert min(R) * 1.05 < max(R), R
    cap = (sorted(R)[K // 2 - 1] + sorted(R)[K // 2]) // 2           # half of the frames fit
    fits = [r <= cap for r in R]
    assert any(fits) and not all(fits)
    # reference: the fitting frames alone, as one batch each, accumulated in frame order
    flat_ref = torch.zeros_like(big.grad_flat)
    first = True
    imgs = {}
    for f in range(K):
        if not fits[f]:
            continue
        e1 = SkinnedFramesEngine(N, J, W, H, 16, 1, dev, cap, grad_flat=flat_ref)
        e1.set_camera(one[0]); e1.set_frames(ins["xyz"], None, ins["w"], A[f:f + 1].contiguous(), ins["smpl_scale"], transl[f:f + 1].contiguous())
        assert e1.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True) == [R[f]]
        e1.backward(ins["sh"], ins["op"], ins["sc"], dL[f:f + 1].contiguous(), accumulate=not first)
        first = False
        torch.cuda.synchronize()
        imgs[f] = e1.color[0].clone()
    eng = SkinnedFramesEngine(N, J, W, H, 16, K, dev, cap)
    for buf in (eng.geom, eng.img, eng.bwd_ws):
        buf.fill_(0xA5)                                               # stale contents must never be followed
    eng.binning.view(K, -1)[:, eng.L.bin_ranges:].fill_(0xA5)
    eng.set_camera(one[0]); eng.set_frames(ins["xyz"], None, ins["w"], A, ins["smpl_scale"], transl)
    Rs = eng.forward(ins["sh"], ins["op"], ins["sc"], sync_num_rendered=True)
    eng.backward(ins["sh"], ins["op"], ins["sc"], dL)
    torch.cuda.synchronize()
    assert Rs == R                                                  # every frame reports its true pair count
    bg = torch.from_numpy(np.ascontiguousarray(s["bg"])).to(dev)
    for f in range(K):
        if fits[f]:
            assert torch.equal(eng.color[f], imgs[f]), f
        else:
            assert torch.equal(eng.color[f], bg[:, None, None].expand(3, H, W)), f
            assert float(eng.d_means2D[f].abs().max()) == 0.0 and float(eng.d_A[f].abs().max()) == 0.0, f
    assert torch.isfinite(eng.grad_flat).all()
    assert torch.equal(eng.grad_flat, flat_ref)


def test_two_batches_of_a_step_share_one_gradient_buffer():
    """accumulate=1 on the K-frame call: a step of 6 frames as batches of 4 + 2 into one buffer == 6 single-frame calls."""
    from sings_amd.engine import SkinnedEngine, SkinnedFramesEngine
    dev = _dev()
    N, J, W, H, K = 12000, 24, 128, 224, 6
    s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 5, K, False, False)
    cap = 24 * N
    ref = SkinnedEngine(N, J, W, H, 16, dev, cap); ref.throughput = True
    ref.set_camera(one[0])
    for f in range(K):
        ref.set_frame(ins["xyz"], None, ins["w"], A[f], ins["smpl_scale"], transl[f])
        ref.forward(ins["sh"], ins["op"], ins["sc"])
        ref._chain = (lambda f=f: (f > 0, None, None))
        ref.backward(ins["sh"], ins["op"], ins["sc"], dL[f])
    torch.cuda.synchronize()
    flat = torch.empty_like(ref.grad_flat)
    a = SkinnedFramesEngine(N, J, W, H, 16, 4, dev, cap, grad_flat=flat)
    b = SkinnedFramesEngine(N, J, W, H, 16, 2, dev, cap, grad_flat=flat)
    for e, lo, hi, acc in ((a, 0, 4, False), (b, 4, 6, True)):
        e.set_camera(one[0])
        e.set_frames(ins["xyz"], None, ins["w"], A[lo:hi].contiguous(), ins["smpl_scale"], transl[lo:hi].contiguous())
        e.forward(ins["sh"], ins["op"], ins["sc"])
        e.backward(ins["sh"], ins["op"], ins["sc"], dL[lo:hi].contiguous(), accumulate=acc)
    torch.cuda.synchronize()
    assert torch.equal(flat, ref.grad_flat)


@pytest.mark.parametrize("K,deg,W,H", [(1, 3, 640, 368), (4, 3, 640, 368), (8, 1, 1600, 1056), (3, 0, 160, 96)])
def test_raster_frames_equal_single_camera_calls_bit_for_bit(K, deg, W, H):
    """Un-skinned path: K cameras of the same Gaussians in one call -- many tiles (the dense composite kernels) and few tiles (the
    deep / sparse pair) -- against K RasterEngine calls that share one gradient buffer."""
    from sings_amd.engine import RasterEngine, RasterFramesEngine
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    N = 30000
    s = synthetic_scene(N, W, H, deg, 7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    views = np.repeat(s["viewmatrix"][None], K, 0).copy(); views[:, 3, 0] = 0.05 * np.arange(K)
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])

    def settings(vm, pm, cp):
        return GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                             scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=deg, campos=cp,
                                             prefiltered=False, debug=False)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    rs = np.random.RandomState(1)
    dL = t(rs.normal(0, 1, (K, 3, H, W)).astype(np.float32))
    cap = 12 * N
    ref = RasterEngine(N, W, H, 16, dev, cap); ref.throughput = True
    per = []
    for f in range(K):
        ref.set_camera(settings(t(views[f]), t(projs[f]), t(cps[f])))
        R = ref.forward(*ins, sync_num_rendered=True)
        ref._chain = (lambda f=f: (f > 0, None, None))
        ref.backward(*ins, dL[f])
        torch.cuda.synchronize()
        per.append((ref.color.clone(), ref.radii.clone(), ref.d_means2D.clone(), R))
    eng = RasterFramesEngine(N, W, H, 16, K, dev, cap)
    eng.set_camera(settings(t(views), t(projs), t(cps)))
    for rep in range(2):
        Rs = eng.forward(*ins, sync_num_rendered=True)
        eng.backward(*ins, dL)
        torch.cuda.synchronize()
        assert Rs == [p[3] for p in per]
        for f in range(K):
            assert torch.equal(eng.color[f], per[f][0]) and torch.equal(eng.radii[f], per[f][1]) and torch.equal(eng.d_means2D[f], per[f][2]), f
        assert torch.equal(eng.grad_flat, ref.grad_flat)


@pytest.mark.parametrize("deg,W,H", [(3, 640, 368), (2, 320, 192), (0, 160, 96)])
def test_raster_two_batches_of_a_step_share_one_gradient_buffer(deg, W, H):
    """accumulate=1 on the K-camera call (the per-Gaussian backward preloads the buffer, then adds its frames in order; dL/dsh sums
    in LDS): 5 cameras as batches of 3 + 2 into one buffer == 5 single-camera calls chained with accumulate."""
    from sings_amd.engine import RasterEngine, RasterFramesEngine
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    N, K = 20000, 5
    s = synthetic_scene(N, W, H, deg, 11)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    views = np.repeat(s["viewmatrix"][None], K, 0).copy(); views[:, 3, 0] = 0.04 * np.arange(K)
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])

    def settings(vm, pm, cp):
        return GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                             scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=deg, campos=cp,
                                             prefiltered=False, debug=False)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(np.random.RandomState(2).normal(0, 1, (K, 3, H, W)).astype(np.float32))
    cap = 12 * N
    ref = RasterEngine(N, W, H, 16, dev, cap); ref.throughput = True
    for f in range(K):
        ref.set_camera(settings(t(views[f]), t(projs[f]), t(cps[f])))
        ref.forward(*ins)
        ref._chain = (lambda f=f: (f > 0, None, None))
        ref.backward(*ins, dL[f])
    torch.cuda.synchronize()
    flat = torch.full_like(ref.grad_flat, float("nan"))
    a = RasterFramesEngine(N, W, H, 16, 3, dev, cap, grad_flat=flat)
    b = RasterFramesEngine(N, W, H, 16, 2, dev, cap, grad_flat=flat)
    for e, lo, hi, acc in ((a, 0, 3, False), (b, 3, 5, True)):
        e.set_camera(settings(t(views[lo:hi]), t(projs[lo:hi]), t(cps[lo:hi])))
        e.forward(*ins)
        e.backward(*ins, dL[lo:hi].contiguous(), accumulate=acc)
    torch.cuda.synchronize()
    assert torch.equal(flat, ref.grad_flat)
    assert float(flat.abs().max()) > 0


def test_a_step_of_four_launches_on_two_streams_sums_like_twelve_single_views():
    """What bench.py does with more launches than streams: ViewBatch deals 4 launches of 3 cameras to 2 streams, one gradient row per
    stream -- the second launch of a stream ADDS to its row (accumulate, decided at backward time) -- and the fold sums the rows.
    The result must be the sum of the 12 views' gradients: rows are compared bit for bit with single-camera chains in the same
    order, the folded sum with their fold."""
    from sings_amd.engine import RasterEngine, RasterFramesEngine, ViewBatch
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    N, W, H, deg, K, B, S = 15000, 320, 192, 3, 3, 4, 2
    s = synthetic_scene(N, W, H, deg, 5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    V = K * B
    views = np.repeat(s["viewmatrix"][None], V, 0).copy(); views[:, 3, 0] = 0.01 * np.arange(V)
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])

    def settings(vm, pm, cp):
        return GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                             scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=deg, campos=cp,
                                             prefiltered=False, debug=False)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(np.random.RandomState(3).normal(0, 1, (V, 3, H, W)).astype(np.float32))
    cap = 12 * N
    per_view = N * (3 + 3 + 4 + 1 + 3 * 16)
    grads = ViewBatch.gradient_rows(S, per_view, dev)
    grads.fill_(float("nan"))
    engs = []
    for b in range(B):
        e = RasterFramesEngine(N, W, H, 16, K, dev, cap, grad_flat=grads[b % S])
        lo = b * K
        e.set_camera(settings(t(views[lo:lo + K]), t(projs[lo:lo + K]), t(cps[lo:lo + K])))
        engs.append(e)
    batch = ViewBatch(engs, grads, S)

    def one(b, e):
        e.forward(*ins)
        e.backward(*ins, dL[b * K:(b + 1) * K].contiguous())
    for rep in range(2):                                            # twice: nothing of the first step may leak into the second
        acc = batch.run(one).clone()
        torch.cuda.synchronize()
    # reference: per stream row, the views of its launches in launch order, chained with accumulate
    ref_rows = []
    for r in range(S):
        ref = RasterEngine(N, W, H, 16, dev, cap); ref.throughput = True
        first = True
        for b in range(r, B, S):
            for f in range(K):
                v = b * K + f
                ref.set_camera(settings(t(views[v]), t(projs[v]), t(cps[v])))
                ref.forward(*ins)
                ref._chain = (lambda first=first: (not first, None, None))
                ref.backward(*ins, dL[v])
                first = False
        torch.cuda.synchronize()
        ref_rows.append(ref.grad_flat.clone())
        assert torch.equal(grads[r], ref_rows[r]), f"row {r}"
    assert torch.equal(acc, ref_rows[0] + ref_rows[1])
    assert torch.isfinite(acc).all() and float(acc.abs().max()) > 0


def test_posed_side_outputs_of_a_k_frame_forward():
    """posed_xyz / posed_rotq / posed_scales [K,P,.] of sg_skinned_forward_frames (what the unfused callers of row a8 read back:
    sings_hybrid.py:400-419) equal the single-frame call's, frame by frame, for isotropic and 6-D-rotation avatars."""
    import ctypes as C
    from sings_amd import _lib
    from sings_amd.engine import SkinnedFramesEngine
    from sings_amd.rasterizer import _ptr
    dev = _dev()
    lib = _lib.load()
    N, J, W, H, K = 9000, 52, 128, 224, 4
    for with_rot in (False, True):
        s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 13, K, False, with_rot)
        cap = 24 * N

        def posed(k, lo):
            e = SkinnedFramesEngine(N, J, W, H, 16, k, dev, cap, with_rot=with_rot, rot_width=6)
            e.set_camera(one[0]); e.set_frames(ins["xyz"], ins["rot"], ins["w"], A[lo:lo + k].contiguous(), ins["smpl_scale"], transl[lo:lo + k].contiguous())
            px = torch.full((k, N, 3), float("nan"), device=dev); pq = torch.full((k, N, 4), float("nan"), device=dev)
            ps = torch.full((k, N, 3), float("nan"), device=dev)
            e._s.flags = e._flags()
            _lib.check(lib.sg_skinned_forward_frames(C.byref(e._s), C.byref(e._fb), N, C.byref(e._k), _ptr(ins["sh"]), _ptr(ins["op"]), _ptr(ins["sc"]),
                                                     _ptr(e.geom), _ptr(e.binning), cap, _ptr(e.img), _ptr(e.color), _ptr(e.radii), _ptr(px), _ptr(pq),
                                                     _ptr(ps), None, e._stream()), "forward with posed outputs")
            torch.cuda.synchronize()
            return px, pq, ps
        px, pq, ps = posed(K, 0)
        assert torch.isfinite(px).all() and torch.isfinite(pq).all() and torch.isfinite(ps).all()
        for f in range(K):
            x1, q1, s1 = posed(1, f)
            assert torch.equal(px[f], x1[0]) and torch.equal(pq[f], q1[0]) and torch.equal(ps[f], s1[0]), (with_rot, f)
        assert float((px[0] - px[K - 1]).abs().max()) > 1e-3


def test_frames_through_the_c_abi_with_precomputed_colours_and_covariances():
    """colors_precomp + cov3D_precomp (the op surface's two optional inputs) through the K-frame entry points
    themselves -- no engine class wraps these: 3 cameras in one call against three K = 1 calls chained with accumulate, bit for
    bit, gradients of the colours and the covariances included."""
    import ctypes as C
    from oracle import raster_oracle as ro
    from sings_amd import _lib
    from sings_amd.engine import _frame_batch
    from sings_amd.rasterizer import GaussianRasterizationSettings, _ptr, _settings_struct
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    lib = _lib.load()
    N, W, H, K = 6000, 256, 160, 3
    s = synthetic_scene(N, W, H, 0, 21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    colors = np.random.RandomState(1).uniform(0, 1, (N, 3)).astype(np.float32)
    o = ro.forward(s["means3D"], s["opacities"], s["viewmatrix"], s["projmatrix"], s["campos"], W, H, s["tanfovx"], s["tanfovy"], s["bg"],
                   colors_precomp=colors, scales=s["scales"], rotations=s["rotations"])
    means, op, col, cov = t(s["means3D"]), t(s["opacities"]), t(colors), t(o["cov3D"])
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    views = np.repeat(s["viewmatrix"][None], K, 0).copy(); views[:, 3, 0] = 0.03 * np.arange(K)
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])
    dL = t(np.random.RandomState(2).normal(0, 1, (K, 3, H, W)).astype(np.float32))
    cap = 16 * N
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run(k, lo, grads, accumulate):
        """cameras lo .. lo + k of the batch in ONE call; returns (images, radii, dL_dmeans2D)"""
        keep = []
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                           scale_modifier=1.0, viewmatrix=t(views[lo:lo + k]), projmatrix=t(projs[lo:lo + k]),
                                           sh_degree=0, campos=t(cps[lo:lo + k]), prefiltered=False, debug=False)
        st = _settings_struct(rs, dev, 0, keep)
        fb = _frame_batch(k, 1, 0)
        L = _lib.SgLayout(); sizes = [C.c_size_t() for _ in range(4)]
        _lib.check(lib.sg_frames_layout(N, W, H, cap, k, C.byref(L), *[C.byref(x) for x in sizes]), "layout")
        u8 = dict(dtype=torch.uint8, device=dev)
        geom, binning, img, bwd = (torch.zeros(x.value, **u8) for x in sizes)
        color = torch.empty((k, 3, H, W), device=dev); radii = torch.empty((k, N), dtype=torch.int32, device=dev)
        m2 = torch.empty((k, N, 3), device=dev)
        nr = (C.c_int64 * k)()
        _lib.check(lib.sg_rasterize_forward_frames(C.byref(st), C.byref(fb), N, _ptr(means), None, _ptr(col), _ptr(op), None, None, _ptr(cov),
                                                   _ptr(geom), _ptr(binning), cap, _ptr(img), _ptr(color), _ptr(radii), nr, stream), "fwd")
        assert all(0 < int(v) <= cap for v in nr)
        _lib.check(lib.sg_rasterize_backward_records_frames(C.byref(st), C.byref(fb), N, _ptr(geom), _ptr(binning), cap, _ptr(img), _ptr(bwd),
                                                            _ptr(dL[lo:lo + k].contiguous()), stream), "bwd records")
        _lib.check(lib.sg_rasterize_backward_gaussians_frames(
            C.byref(st), C.byref(fb), N, _ptr(means), None, _ptr(col), _ptr(op), None, None, _ptr(cov), _ptr(radii), _ptr(geom),
            _ptr(binning), cap, _ptr(bwd), int(accumulate), _ptr(grads["means"]), _ptr(m2), None, _ptr(grads["col"]), _ptr(grads["op"]),
            None, None, _ptr(grads["cov"]), stream), "bwd gaussians")
        torch.cuda.synchronize()
        return color, radii, m2

    def buffers():
        nan = float("nan")
        return dict(means=torch.full((N, 3), nan, device=dev), col=torch.full((N, 3), nan, device=dev),
                    op=torch.full((N, 1), nan, device=dev), cov=torch.full((N, 6), nan, device=dev))
    ref = buffers()
    singles = [run(1, f, ref, f > 0) for f in range(K)]
    got = buffers()
    color, radii, m2 = run(K, 0, got, False)
    for f in range(K):
        assert torch.equal(color[f], singles[f][0][0]) and torch.equal(radii[f], singles[f][1][0]) and torch.equal(m2[f], singles[f][2][0]), f
    for k in ref:
        assert torch.isfinite(got[k]).all() and torch.equal(got[k], ref[k]), k
    assert float(got["cov"].abs().max()) > 0 and float(got["col"].abs().max()) > 0
    # and the single camera 0 agrees with the oracle's image (the precomputed inputs really are what was rendered)
    border = o["margin"] < 2e-5 if "margin" in o else np.zeros((H, W), bool)
    assert np.abs(singles[0][0][0].cpu().numpy() - o["color"]).max(0)[~border].max() <= 1e-5


def test_photo_loss_frames_equal_single_calls():
    from sings_amd.photo_loss import PhotoLossEngine
    dev = _dev()
    K, W, H = 5, 200, 136
    g = torch.Generator(device="cpu").manual_seed(3)
    raw = (torch.rand(K, 3, H, W, generator=g) * 1.4 - 0.2).to(dev)
    gt = torch.rand(K, 3, H, W, generator=g).to(dev)
    mask = (torch.rand(K, H, W, generator=g) > 0.3).float().to(dev)
    bg = torch.tensor([1.0, 0.5, 0.25], device=dev)
    one = PhotoLossEngine(W, H, dev)
    many = PhotoLossEngine(W, H, dev, K=K)
    grad = many(raw, gt, mask, bg)
    torch.cuda.synchronize()
    for f in range(K):
        gf = one(raw[f], gt[f], mask[f], bg)
        torch.cuda.synchronize()
        assert torch.equal(grad[f], gf) and torch.equal(many.losses[f], one.losses), f
    # one target / one mask for all frames
    shared = PhotoLossEngine(W, H, dev, K=K)
    g2 = shared(raw, gt[0], mask[0], bg)
    torch.cuda.synchronize()
    for f in range(K):
        gf = one(raw[f], gt[0], mask[0], bg)
        torch.cuda.synchronize()
        assert torch.equal(g2[f], gf) and torch.equal(shared.losses[f], one.losses), f


@pytest.mark.parametrize("deg", [2, 3])
def test_planar_sh_gradients_of_the_k_camera_raster_call(deg):
    """The same flag on the un-skinned K-camera call (``RasterFramesEngine(sh_planar=True)``): (deg+1)^2 planes written, equal to the
    default layout's rows transposed, with and without accumulate (the planar preload of the per-Gaussian backward)."""
    from sings_amd.engine import RasterFramesEngine
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    N, W, H, K = 12000, 320, 192, 3
    s = synthetic_scene(N, W, H, deg, 17)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    views = np.repeat(s["viewmatrix"][None], K, 0).copy(); views[:, 3, 0] = 0.02 * np.arange(K)
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(views), projmatrix=t(projs), sh_degree=deg, campos=t(cps),
                                       prefiltered=False, debug=False)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    dL = t(np.random.RandomState(4).normal(0, 1, (K, 3, H, W)).astype(np.float32))
    nc = (deg + 1) ** 2

    def run(planar, accumulate):
        e = RasterFramesEngine(N, W, H, 16, K, dev, 12 * N, sh_planar=planar)
        e.set_camera(rs)
        e.grad_flat.fill_(0.5 if accumulate else float("nan"))
        e.forward(*ins)
        e.backward(*ins, dL, accumulate=accumulate)
        torch.cuda.synchronize()
        return e
    for accumulate in (False, True):
        a, b = run(False, accumulate), run(True, accumulate)
        assert b.active_floats(deg) == N * (11 + 3 * nc) and a.active_floats(deg) == N * 59
        for x, y in ((a.d_means3D, b.d_means3D), (a.d_scales, b.d_scales), (a.d_rots, b.d_rots), (a.d_opacity, b.d_opacity)):
            assert torch.equal(x, y)
        assert torch.equal(a.d_sh[:, :nc, :].permute(1, 0, 2), b.d_sh[:nc]) and float(b.d_sh[:nc].abs().max()) > 0
        rest = b.d_sh[nc:]
        assert bool((rest == 0.5).all()) if accumulate else bool(torch.isnan(rest).all())       # planes not in use: never touched


def test_frames_api_rejects_bad_batches():
    from sings_amd.engine import SkinnedFramesEngine
    dev = _dev()
    with pytest.raises(ValueError):
        SkinnedFramesEngine(100, 24, 64, 64, 16, 17, dev, 4096)
    with pytest.raises(ValueError):
        SkinnedFramesEngine(100, 24, 64, 64, 16, 0, dev, 4096)


@pytest.mark.parametrize("K", [1, 4])
def test_planar_sh_gradients_are_the_reference_layout_transposed(K):
    """SG_FLAG_SH_PLANAR (``sh_planar=True``): dL/dsh coefficient-major [M,P,3], only the (sh_degree+1)^2 planes in use written --
    the gradient of a step is the prefix ``grad_flat[:active_floats(deg)]`` (10 of 55 floats per Gaussian at degree 0).  Every
    gradient equals the default layout's, bit for bit (the planes in use = the rows in use transposed; the rest of the SH block
    is never touched), for the single-frame kernel, the K-frame kernel and their accumulate variants."""
    from sings_amd.engine import SkinnedEngine, SkinnedFramesEngine
    dev = _dev()
    N, J, W, H = 9000, 24, 128, 224
    s, ins, A, transl, one, stacked, dL = _avatar(N, J, W, H, 9, K, False, False)
    cap = 24 * N

    def run(planar, accumulate):
        if K == 1:
            e = SkinnedEngine(N, J, W, H, 16, dev, cap, sh_planar=planar); e.throughput = True
            e.set_camera(one[0]); e.set_frame(ins["xyz"], None, ins["w"], A[0], ins["smpl_scale"], transl[0])
            e.grad_flat.fill_(0.25 if accumulate else 0.0)
            if planar and accumulate:
                e.d_sh[1:].zero_()                                   # (planes not in use: never written, must come zeroed)
            e.forward(ins["sh"], ins["op"], ins["sc"])
            e._chain = lambda: (accumulate, None, None)
            e.backward(ins["sh"], ins["op"], ins["sc"], dL[0])
        else:
            e = SkinnedFramesEngine(N, J, W, H, 16, K, dev, cap, sh_planar=planar)
            e.set_camera(one[0]); e.set_frames(ins["xyz"], None, ins["w"], A, ins["smpl_scale"], transl)
            e.grad_flat.fill_(0.25 if accumulate else 0.0)
            if planar and accumulate:
                e.d_sh[1:].zero_()
            e.forward(ins["sh"], ins["op"], ins["sc"])
            e.backward(ins["sh"], ins["op"], ins["sc"], dL, accumulate=accumulate)
        torch.cuda.synchronize()
        return e

    for accumulate in (False, True):
        a, b = run(False, accumulate), run(True, accumulate)
        assert b.active_floats(0) == N * 10 and a.active_floats(0) == N * 55
        assert torch.equal(a.d_xyz, b.d_xyz) and torch.equal(a.d_scales, b.d_scales) and torch.equal(a.d_opacity, b.d_opacity)
        assert torch.equal(a.d_sh[:, 0, :], b.d_sh[0]) and float(b.d_sh[0].abs().max()) > 0
        assert not bool(b.d_sh[1:].any())                            # never touched
        assert torch.equal(b.grad_flat[:N * 7], a.grad_flat[:N * 7])
        assert not bool(b.grad_flat[b.active_floats(0):].any())


def test_cfg3_eight_cameras_in_one_launch_against_the_oracle():
    """ONE K = 8 launch per kernel of BASELINE configs[2] at FULL size (200 k Gaussians, 1080p, SH degree 3: the launch the
    headline bench line times, sg_rasterize_forward_frames / *_backward_*_frames with bench.py's cameras) compared with the CPU
    oracle DIRECTLY, not through "K-camera call == K single-camera calls == oracle": every camera's pair count and radii bit for
    bit; cameras 0 and 7: sorted lists and ranges bit for bit, image (<= 1e-5 off borderline pixels, those within the oracle's
    flip bound), the screen-space gradient; and the gradient row the launch leaves behind = the SUM of the oracle's gradients
    over the eight cameras (means, scales, rotations, opacity, SH).  ~35 s of scalar oracle."""
    from oracle import raster_oracle as ro
    from sings_amd.engine import RasterFramesEngine
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.scene import synthetic_scene
    dev = _dev()
    N, W, H, deg, K = 200000, 1920, 1080, 3, 8
    s = synthetic_scene(N, W, H, deg, 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P_T = np.linalg.inv(s["viewmatrix"]) @ s["projmatrix"]
    views = np.repeat(s["viewmatrix"][None], K, 0).copy(); views[:, 3, 0] = 0.012 * np.arange(K)       # bench.py's cameras 0..7
    projs = np.stack([(v @ P_T).astype(np.float32) for v in views])
    cps = np.stack([np.linalg.inv(v)[3, :3].astype(np.float32) for v in views])
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], bg=t(s["bg"]),
                                       scale_modifier=1.0, viewmatrix=t(views), projmatrix=t(projs), sh_degree=deg, campos=t(cps),
                                       prefiltered=False, debug=False)
    ins = [t(s[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    eng = RasterFramesEngine(N, W, H, 16, K, dev, 5 * N)
    eng.set_camera(rs)
    Rs = eng.forward(*ins, sync_num_rendered=True)
    assert max(Rs) <= eng.cap
    color = eng.color.cpu().numpy(); radii = eng.radii.cpu().numpy()
    L, Tn = eng.L, ((W + 15) // 16) * ((H + 15) // 16)
    bins = eng.binning.view(K, -1)
    dL_n = np.random.RandomState(12).normal(0, 1, (K, 3, H, W)).astype(np.float32)
    sums = {k: 0.0 for k in ("dL_dmeans3D", "dL_dscales", "dL_drots", "dL_dopacity", "dL_dsh")}
    keep = {}
    BORDER = 2e-5
    for f in range(K):
        o = ro.forward(s["means3D"], s["opacities"], views[f], projs[f], cps[f], W, H, s["tanfovx"], s["tanfovy"], s["bg"],
                       scales=s["scales"], rotations=s["rotations"], shs=s["shs"], sh_degree=deg)
        assert Rs[f] == o["R"] and np.array_equal(radii[f], o["radii"]), (f, Rs[f], o["R"])
        border = o["margin"] < BORDER
        dL_n[f][:, border] = 0
        g = ro.backward(o, dL_n[f])
        for k in sums:
            sums[k] = sums[k] + g[k].astype(np.float64)
        if f in (0, K - 1):
            ranges = bins[f, L.bin_ranges:L.bin_ranges + 8 * Tn].view(torch.int32).view(Tn, 2).cpu().numpy()
            plist = bins[f, L.bin_point_list:L.bin_point_list + 4 * Rs[f]].view(torch.int32).cpu().numpy()
            assert np.array_equal(ranges.astype(np.uint32), o["ranges"]) and np.array_equal(plist.astype(np.uint32), o["point_list"]), f
            diff = np.abs(color[f] - o["color"]).max(0)
            assert diff[~border].max() <= 1e-5, (f, diff[~border].max())
            assert border.sum() == 0 or (diff[border] - (1e-5 + 1.001 * o["flip"][border])).max() <= 0, f
            keep[f] = g["dL_dmean2D"]
    eng.backward(*ins, t(dL_n))
    torch.cuda.synchronize()

    def close(name, a, b, rtol=2e-4, atol=2e-6):
        a = np.asarray(a, np.float64).reshape(b.shape); b = np.asarray(b, np.float64)
        scale = np.abs(b).max() + 1e-30
        bad = np.abs(a - b) > rtol * np.abs(b) + atol * scale
        assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} off; worst {np.abs(a - b).max():.3e} (scale {scale:.3e})"
    for f, g2 in keep.items():
        close(f"means2D[{f}]", eng.d_means2D[f].cpu().numpy(), g2)
    for name, ten, key in (("means3D", eng.d_means3D, "dL_dmeans3D"), ("scales", eng.d_scales, "dL_dscales"),
                           ("rotations", eng.d_rots, "dL_drots"), ("opacity", eng.d_opacity, "dL_dopacity"), ("sh", eng.d_sh, "dL_dsh")):
        close("sum " + name, ten.cpu().numpy(), sums[key])
