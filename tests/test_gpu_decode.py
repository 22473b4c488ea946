"""GPU parity of the attribute decode (tri-plane kernels + bias/activation kernels through the C ABI, library GEMMs)
with golden vectors produced by the reference's modules and with the CPU oracle at the shipped plane configuration.
Tolerances: fp32 with a different summation order (GEMM blocking, atomics) -> 2e-5 relative + 2e-6 of the scale."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_oracle as do

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_golden.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(a, b, rtol=2e-5, atol_scale=2e-6, atol=0.0, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b)
    assert (err <= rtol * np.abs(b) + atol_scale * scale + atol).all(), (what, err.max(), scale)


def _field(dev):
    from sings_amd.decode import HexPlaneField
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32,
           'resolution': [int(r) for r in G["tp_res"]], 'multires': [int(m) for m in G["tp_multires"]]}
    f = HexPlaneField(cfg, device=dev)
    sd = {f"grids.{s}.{c}": torch.from_numpy(G[f"tp_plane_{s}_{c}"]) for s in range(len(cfg['multires'])) for c in range(3)}
    f.set_aabb(G["tp_aabb"][0].tolist(), G["tp_aabb"][1].tolist())      # (makes aabb a state_dict entry, as in the reference)
    sd["aabb"] = torch.from_numpy(G["tp_aabb"])
    f.load_state_dict(sd)                                     # the reference's parameter names
    return f.to(dev)


def test_triplane_golden_forward_backward():
    dev = _dev()
    f = _field(dev)
    pts = torch.from_numpy(G["tp_pts"]).to(dev).requires_grad_(True)
    feats = f(pts)
    _close(feats.detach().cpu().numpy(), G["tp_feats"])
    (feats * torch.from_numpy(G["tp_w"]).to(dev)).sum().backward()
    _close(pts.grad.cpu().numpy(), G["tp_dpts"])
    for s, gp in enumerate(f.grids):
        for c, p in enumerate(gp):
            _close(p.grad.cpu().numpy(), G[f"tp_dplane_{s}_{c}"])


def test_decoders_golden_forward_backward():
    from sings_amd.decode import AppearanceDecoder, GeometryDecoder
    dev = _dev()
    N = G["geo_iso_x"].shape[0]
    for tag, iso in (("iso", True), ("aniso", False)):
        g = GeometryDecoder(n_features=64, isotropic=iso)
        g.load_state_dict({k[len(f"geo_{tag}_p_"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"geo_{tag}_p_")})
        g = g.to(dev)
        x = torch.from_numpy(G[f"geo_{tag}_x"]).to(dev).requires_grad_(True)
        o = g(x)
        for k in ('xyz_offsets', 'scales', 'scales_aux') + (() if iso else ('rotations',)):
            _close(o[k].detach().cpu().numpy(), G[f"geo_{tag}_o_{k}"])
        loss = (o['xyz_offsets'] * 1.3).sum() + (o['scales'] ** 2).sum() + o['scales_aux'].sum() * 0.1
        if not iso:
            loss = loss + (o['rotations'] * 0.7).sum()
        loss.backward()
        _close(x.grad.cpu().numpy(), G[f"geo_{tag}_dx"], what=f"geo {tag} dx")
        for k, p in g.named_parameters():
            _close(p.grad.cpu().numpy(), G[f"geo_{tag}_g_{k}"], atol_scale=2e-5, atol=2e-5, what=f"geo {tag} {k}")
            # (parameter gradients are cancelling sums over the N = 700 points of terms of magnitude <= ~1: fp32 sum noise
            #  ~1e-5 absolute in the reference's own result as well)
    a = AppearanceDecoder(n_features=64)
    a.load_state_dict({k[len("app_p_"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("app_p_")})
    a = a.to(dev)
    x = torch.from_numpy(G["geo_aniso_x"]).to(dev).requires_grad_(True)
    a.reset_opacity(x.detach())
    _close(a.opacity_offset.cpu().numpy(), G["app_offset"], atol_scale=5e-6)
    a.opacity_offset = torch.from_numpy(G["app_offset"]).to(dev)        # identical offsets for the comparison below
    o = a(x)
    _close(o['shs'].detach().cpu().numpy(), G["app_o_shs"])
    _close(o['opacity'].detach().cpu().numpy(), G["app_o_opacity"])
    ((o['shs'] ** 2).sum() * 0.5 + (o['opacity'] * torch.linspace(-1, 1, N, device=dev)[:, None]).sum()).backward()
    _close(x.grad.cpu().numpy(), G["app_dx"])
    for k, p in a.named_parameters():
        _close(p.grad.cpu().numpy(), G[f"app_g_{k}"], atol_scale=2e-5, atol=2e-5, what=f"app {k}")


def test_shipped_config_vs_oracle_and_decode_attributes():
    """kplanes of human_complex.yaml (32 features, 64^3, multires 1/2/4), 20 k points: features and all gradients vs the
    CPU oracle; decode_attributes returns the keys of SinGS.get_gs_attrs."""
    from sings_amd.decode import AppearanceDecoder, GeometryDecoder, HexPlaneField, decode_attributes
    dev = _dev()
    torch.manual_seed(0)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [64, 64, 64], 'multires': [1, 2, 4]}
    f = HexPlaneField(cfg, device=dev)
    N = 20000
    pts_c = (torch.rand(N, 3) * 1.9 - 0.95)
    w_c = torch.randn(N, 96)
    pts = pts_c.to(dev).requires_grad_(True)
    feats = f(pts)
    (feats * w_c.to(dev)).sum().backward()
    grids_c = [[p.detach().cpu().clone().requires_grad_(True) for p in gp] for gp in f.grids]
    pc = pts_c.clone().requires_grad_(True)
    fo = do.triplane_features(pc, grids_c, f.aabb.detach().cpu())
    (fo * w_c).sum().backward()
    _close(feats.detach().cpu().numpy(), fo.detach().numpy())
    _close(pts.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol_scale=1e-5)
    for gp, gc in zip(f.grids, grids_c):
        for p, q in zip(gp, gc):
            _close(p.grad.cpu().numpy(), q.grad.numpy(), rtol=1e-4, atol_scale=1e-5)
    g = GeometryDecoder(96).to(dev); a = AppearanceDecoder(96).to(dev)
    out = decode_attributes(pts.detach(), f, g, a, thickness_factor=0.5, scaling_multiplier=torch.full((N, 1), 2.0, device=dev))
    assert set(out) == {"xyz_canon", "xyz_offsets", "rot6d_canon", "scales_aux", "scales", "opacity", "shs"}
    assert out["shs"].shape == (N, 16, 3) and out["opacity"].shape == (N, 1) and out["scales"].shape == (N, 3)
    sd_g = {k: v.detach().cpu() for k, v in g.state_dict().items()}
    og = do.geometry_decoder(fo.detach(), sd_g)
    ref_scales = og['scales'].clone(); ref_scales[:, -1] *= 0.5; ref_scales = ref_scales * 2.0
    _close(out["scales"].detach().cpu().numpy(), ref_scales.numpy(), rtol=1e-4, atol_scale=1e-5)
    _close(out["xyz_canon"].detach().cpu().numpy(), (pts_c + og['xyz_offsets']).numpy(), rtol=1e-4, atol_scale=1e-5)


def test_triplane_backward_clustered_and_border_points():
    """The plane gradient is a sort + segmented scatter: a cloud that piles thousands of points on a few texel cells
    (an avatar in its bounding box), exact duplicates, points on / beyond the box faces and on texel-cell boundaries of
    every level, N not a multiple of the 32-point runs -- against the CPU oracle.  Non-cubic resolution."""
    from sings_amd.decode import HexPlaneField
    dev = _dev()
    torch.manual_seed(3)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [24, 16, 20], 'multires': [1, 2, 4]}
    f = HexPlaneField(cfg, bounds=1.0, device=dev)
    blob = torch.randn(6000, 3) * 0.02 + torch.tensor([0.31, -0.12, 0.05])
    dup = blob[:500].clone()
    faces = torch.rand(300, 3) * 2 - 1
    faces[:100, 0] = 1.0; faces[100:200, 1] = -1.0; faces[200:250, 2] = 1.3; faces[250:, 0] = -1.7
    lattice = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 24), torch.linspace(-1, 1, 31), torch.linspace(-1, 1, 5),
                                         indexing="ij"), -1).reshape(-1, 3)       # exact texel coordinates of level 0 in x
    pts_c = torch.cat([blob, dup, faces, lattice, torch.rand(1237, 3) * 2 - 1])
    N = pts_c.shape[0]
    assert N % 32 != 0
    w_c = torch.randn(N, 96)
    pts = pts_c.to(dev).requires_grad_(True)
    feats = f(pts)
    (feats * w_c.to(dev)).sum().backward()
    grids_c = [[p.detach().cpu().clone().requires_grad_(True) for p in gp] for gp in f.grids]
    pc = pts_c.clone().requires_grad_(True)
    fo = do.triplane_features(pc, grids_c, f.aabb.detach().cpu())
    (fo * w_c).sum().backward()
    _close(feats.detach().cpu().numpy(), fo.detach().numpy())
    _close(pts.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol_scale=1e-5)
    for gp, gc in zip(f.grids, grids_c):
        for p, q in zip(gp, gc):
            _close(p.grad.cpu().numpy(), q.grad.numpy(), rtol=1e-4, atol_scale=2e-5)


def test_weight_grad_shapes():
    """sg_weight_grad over the shapes the decoders use and the edges of its two code paths (streaming
    dot products up to 12 outputs; more: MFMA tiles): dW = dz^T x, db = column sums, against float64."""
    import ctypes as C
    from sings_amd import _lib
    dev = _dev()
    lib = _lib.load()
    torch.manual_seed(5)
    for N in (1, 37, 10007):
        for Cin in (32, 64, 96, 128):
            for Cout in (1, 2, 3, 4, 5, 6, 9, 12, 13, 48, 128):
                dz = torch.randn(N, Cout, device=dev); x = torch.randn(N, Cin, device=dev)
                dW = torch.empty(Cout, Cin, device=dev); db = torch.empty(Cout, device=dev)
                ws = torch.empty(int(lib.sg_weight_grad_ws_bytes(N, Cout, Cin)), dtype=torch.uint8, device=dev)
                _lib.check(lib.sg_weight_grad(N, Cout, Cin, C.c_void_p(dz.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(ws.data_ptr()),
                                              C.c_void_p(dW.data_ptr()), C.c_void_p(db.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "weight gradient")
                ref = dz.double().t() @ x.double()
                _close(dW.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol_scale=1e-5, what=f"dW {N} {Cout} {Cin}")
                _close(db.cpu().numpy(), dz.double().sum(0).cpu().numpy(), rtol=1e-5, atol_scale=1e-5, what=f"db {N} {Cout} {Cin}")


@pytest.mark.parametrize("res,multires", [([8, 8, 8], [1]), ([8, 6, 10], [1, 2]), ([16, 12, 20], [1, 2, 4, 8]),
                                          ([96, 64, 80], [1, 2, 4, 8]), ([5, 7, 3], [4, 1])])
def test_triplane_plane_configurations(res, multires):
    """The plane gradient sorts the points by (finest cell, parity of the cell on every coarser level): one level (no
    sub-key), four levels, non-cubic planes, a finest level that is not the last one, and a key space beyond the 2^24 cap
    (96 x 8 = 768 texels: the coarsest sub-keys are dropped) -- features and all gradients against the CPU oracle."""
    from sings_amd.decode import HexPlaneField
    dev = _dev()
    torch.manual_seed(11)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': res, 'multires': multires}
    f = HexPlaneField(cfg, bounds=1.0, device=dev)
    N = 3001
    pts_c = torch.cat([torch.rand(N - 600, 3) * 2.2 - 1.1, torch.randn(600, 3) * 0.01 + 0.4])
    w_c = torch.randn(N, 32 * len(multires))
    pts = pts_c.to(dev).requires_grad_(True)
    feats = f(pts)
    (feats * w_c.to(dev)).sum().backward()
    grids_c = [[p.detach().cpu().clone().requires_grad_(True) for p in gp] for gp in f.grids]
    pc = pts_c.clone().requires_grad_(True)
    fo = do.triplane_features(pc, grids_c, f.aabb.detach().cpu())
    (fo * w_c).sum().backward()
    _close(feats.detach().cpu().numpy(), fo.detach().numpy())
    _close(pts.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol_scale=1e-5)
    for gp, gc in zip(f.grids, grids_c):
        for p, q in zip(gp, gc):
            _close(p.grad.cpu().numpy(), q.grad.numpy(), rtol=1e-4, atol_scale=2e-5)


@pytest.mark.parametrize("N,Cin,Cout,act", [(1000, 96, 128, 1), (4097, 128, 128, 1), (333, 128, 3, 0), (2050, 128, 6, 0),
                                            (777, 128, 1, 0), (5000, 96, 64, 1), (1234, 64, 64, 1), (999, 64, 48, 0),
                                            (640, 64, 1, 2), (31, 128, 128, 3), (50000, 128, 128, 1),
                                            # the skinny streaming kernel (Cout <= 6, Cin 64 | 128) at the training size, ragged
                                            (150001, 128, 3, 0), (150000, 64, 1, 2), (70003, 128, 6, 0), (13, 64, 4, 0)])
def test_fused_linear_layer_kernels_vs_torch(N, Cin, Cout, act):
    """sg_linear_forward / sg_linear_backward (one MFMA kernel per layer and direction, bias + activation fused) against the
    same layer in torch fp64 on the CPU: h, the saved pre-activation, dx, dW, db."""
    from sings_amd.decode import linear_act
    dev = _dev()
    g = torch.Generator().manual_seed(N + Cin + Cout + act)
    lin = torch.nn.Linear(Cin, Cout)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(Cout, Cin, generator=g) / Cin ** 0.5); lin.bias.copy_(torch.randn(Cout, generator=g))
    x = torch.randn(N, Cin, generator=g)
    ro = torch.randn(N, 1, generator=g) if act == 2 else None
    up = torch.randn(N, Cout, generator=g)
    # reference in fp64
    l64 = torch.nn.Linear(Cin, Cout).double()
    l64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    z = l64(x64)
    if act == 1: h64 = torch.nn.functional.gelu(z)
    elif act == 2: h64 = torch.sigmoid(z + ro.double())
    elif act == 3: h64 = torch.log(torch.exp(z) + 1)
    else: h64 = z
    (h64 * up.double()).sum().backward()
    lin = lin.to(dev)
    xd = x.to(dev).requires_grad_(True)
    h = linear_act(xd, lin, act, None if ro is None else ro.to(dev))
    (h * up.to(dev)).sum().backward()
    _close(h.detach().cpu().numpy(), h64.detach().numpy(), rtol=2e-5, atol_scale=2e-6, what="h")
    _close(xd.grad.cpu().numpy(), x64.grad.numpy(), rtol=2e-5, atol_scale=2e-6, what="dx")
    _close(lin.weight.grad.cpu().numpy(), l64.weight.grad.numpy(), rtol=3e-5, atol_scale=3e-6, what="dW")
    _close(lin.bias.grad.cpu().numpy(), l64.bias.grad.numpy(), rtol=3e-5, atol_scale=3e-6, what="db")


@pytest.mark.parametrize("N,Cin", [(3000, 128), (4099, 64), (777, 96)])
def test_linear_fan_accumulates_input_gradient_like_separate_layers(N, Cin):
    """Several layers on one activation (trunk + heads, decoders.py:75-94): `linear_fan` lets the consumers add their dx into one
    array (sg_linear_backward_accumulate) -- same outputs and gradients as torch's fp64 layers + autograd's sum, ragged N."""
    from sings_amd.decode import ACT_GELU, ACT_NONE, ACT_SIGMOID, linear_fan
    dev = _dev()
    torch.manual_seed(N + Cin)
    lins = [torch.nn.Linear(Cin, 128), torch.nn.Linear(Cin, 3), torch.nn.Linear(Cin, 48), torch.nn.Linear(Cin, 1)]
    acts = [ACT_GELU, ACT_NONE, ACT_NONE, ACT_SIGMOID]
    off = torch.rand(N, 1) * 0.3
    x = torch.randn(N, Cin)
    ups = [torch.randn(N, l.out_features) for l in lins]
    # fp64 reference
    xr = x.double().requires_grad_(True)
    ref_out, loss = [], 0.0
    for l, a, u in zip(lins, acts, ups):
        z = xr @ l.weight.double().T + l.bias.double()
        h = torch.nn.functional.gelu(z) if a == ACT_GELU else (torch.sigmoid(z + off.double()) if a == ACT_SIGMOID else z)
        ref_out.append(h)
        loss = loss + (h * u.double()).sum()
    params = [p for l in lins for p in (l.weight, l.bias)]
    ref_g = torch.autograd.grad(loss, [xr] + params)
    # product
    glins = [torch.nn.Linear(Cin, l.out_features).to(dev) for l in lins]
    for gl, l in zip(glins, lins):
        gl.load_state_dict(l.state_dict())
    xg = x.to(dev).requires_grad_(True)
    outs = linear_fan(xg, [(gl, a, off.to(dev) if a == ACT_SIGMOID else None) for gl, a in zip(glins, acts)])
    lossg = sum((o * u.to(dev)).sum() for o, u in zip(outs, ups))
    gg = torch.autograd.grad(lossg, [xg] + [p for gl in glins for p in (gl.weight, gl.bias)])
    for o, r in zip(outs, ref_out):
        _close(o.detach().cpu().numpy(), r.detach().numpy(), what="fan output")
    for g, r in zip(gg, ref_g):
        _close(g.cpu().numpy(), r.numpy(), rtol=5e-5, atol_scale=5e-6, what="fan gradient")


@pytest.mark.parametrize("N,Cin,couts,acts", [
    (150000, 128, (128, 3, 6), ("gelu", "none", "none")),      # GeometryDecoder's heads (decoders.py:75-94)
    (150000, 64, (48, 1), ("none", "sigmoid")),                # AppearanceDecoder's (decoders.py:41-49)
    (4099, 128, (128, 3), ("gelu", "none")),                   # isotropic: no rotations head; ragged N
    (777, 96, (64, 8, 8), ("gelu", "sigmoid", "none")),        # 16 extra columns, a sigmoid head first
    (33, 32, (20, 1), ("none", "none")),                       # fewer rows than a tile, Cout not a multiple of 8
])
def test_fan_backward_heads_ride_in_the_wide_kernel(N, Cin, couts, acts):
    """sg_linear_backward_fan: the narrow heads' dz columns as extra columns of the wide layer's reduction.  Against fp64 autograd
    (values), and against the per-layer path (`fuse_fan_backward(False)`: one accumulate pass per head) at fp32 round-off."""
    from sings_amd import decode
    from sings_amd.decode import ACT_GELU, ACT_NONE, ACT_SIGMOID, linear_fan
    code = {"gelu": ACT_GELU, "none": ACT_NONE, "sigmoid": ACT_SIGMOID}
    dev = _dev()
    torch.manual_seed(N + Cin + len(couts))
    lins = [torch.nn.Linear(Cin, c) for c in couts]
    A = [code[a] for a in acts]
    off = torch.rand(N, 1) * 0.3                                 # (the opacity offset: a row offset exists for 1-column sigmoid heads)
    with_off = [a == ACT_SIGMOID and c == 1 for a, c in zip(A, couts)]
    x = torch.randn(N, Cin)
    ups = [torch.randn(N, c) for c in couts]
    xr = x.double().requires_grad_(True)
    loss = 0.0
    for l, a, u, wo in zip(lins, A, ups, with_off):
        z = xr @ l.weight.double().T + l.bias.double()
        h = torch.nn.functional.gelu(z) if a == ACT_GELU else (torch.sigmoid(z + (off.double() if wo else 0.0)) if a == ACT_SIGMOID else z)
        loss = loss + (h * u.double()).sum()
    params = [p for l in lins for p in (l.weight, l.bias)]
    ref_g = torch.autograd.grad(loss, [xr] + params)
    glins = [torch.nn.Linear(Cin, c).to(dev) for c in couts]
    for gl, l in zip(glins, lins):
        gl.load_state_dict(l.state_dict())

    def run():
        xg = x.to(dev).requires_grad_(True)
        outs = linear_fan(xg, [(gl, a, off.to(dev) if wo else None) for gl, a, wo in zip(glins, A, with_off)])
        lossg = sum((o * u.to(dev)).sum() for o, u in zip(outs, ups))
        return torch.autograd.grad(lossg, [xg] + [p for gl in glins for p in (gl.weight, gl.bias)])

    fused = run()
    decode.fuse_fan_backward(False)
    try:
        plain = run()
    finally:
        decode.fuse_fan_backward(True)
    for g, r in zip(fused, ref_g):
        _close(g.cpu().numpy(), r.numpy(), rtol=5e-5, atol_scale=5e-6, what="fan gradient (fused)")
    for g, q in zip(fused, plain):
        _close(g.cpu().numpy(), q.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="fused vs per-layer")
    for g, q in zip(fused[1:], plain[1:]):                      # weight / bias gradients: the same kernels on the same dz
        assert torch.equal(g, q) or float((g - q).abs().max()) <= 1e-6 * float(q.abs().max())


@pytest.mark.parametrize("order", ["first_then_second", "second_then_first", "only_second"])
def test_layers_sharing_an_input_collect_one_gradient(order):
    """decode.linear_act_shared: two layers on the same input as two autograd nodes with ONE input gradient -- whichever backward
    runs first writes it, the other adds into that tensor and hands autograd nothing.  Same gradients as `linear_fan` (one node) in
    either execution order, and with only one of the two layers under the loss."""
    from sings_amd import decode
    from sings_amd.decode import ACT_GELU
    dev = _dev()
    torch.manual_seed(31)
    N, Cin = 20011, 96
    l0, l1 = torch.nn.Linear(Cin, 128).to(dev), torch.nn.Linear(Cin, 64).to(dev)
    x0 = torch.randn(N, Cin, device=dev)
    u0, u1 = torch.randn(N, 128, device=dev), torch.randn(N, 64, device=dev)
    params = list(l0.parameters()) + list(l1.parameters())

    def run(shared):
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if shared:
            slot = {}
            h0 = decode.linear_act_shared(x, l0, ACT_GELU, slot)
            h1 = decode.linear_act_shared(x, l1, ACT_GELU, slot)
        else:
            h0, h1 = decode.linear_fan(x, [(l0, ACT_GELU, None), (l1, ACT_GELU, None)])
        a, b = (h0 * u0).sum(), (h1 * u1).sum()
        if order == "only_second":
            b.backward()
        elif order == "first_then_second" or not shared:
            (a + b).backward()                                   # (autograd visits the younger node, h1's, first)
        else:
            # force the other execution order: h0's backward first, then h1's, in ONE pass
            torch.autograd.backward([a, b.detach() * 0 + b], [torch.ones((), device=dev)] * 2)
        torch.cuda.synchronize()
        return [x.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in params]

    ref, got = run(False), run(True)
    for r, g in zip(ref, got):
        if r is None or g is None:                               # (a layer outside the loss: no gradient or a zero gradient)
            assert all(t is None or float(t.abs().max()) == 0.0 for t in (r, g))
            continue
        _close(g.cpu().numpy(), r.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="shared-input layers")


def test_layers_sharing_an_input_over_two_backward_passes():
    """The shared input gradient of decode.linear_act_shared lives for ONE backward pass: a.backward(retain_graph=True) followed by
    b.backward() (each pass reaches one of the two layers) accumulates the same x.grad and parameter gradients as `linear_fan`, and a
    second full pass over the retained graph doubles them instead of losing the input gradient."""
    from sings_amd import decode
    from sings_amd.decode import ACT_GELU
    dev = _dev()
    torch.manual_seed(32)
    N, Cin = 10007, 96
    l0, l1 = torch.nn.Linear(Cin, 128).to(dev), torch.nn.Linear(Cin, 64).to(dev)
    x0 = torch.randn(N, Cin, device=dev)
    u0, u1 = torch.randn(N, 128, device=dev), torch.randn(N, 64, device=dev)
    params = list(l0.parameters()) + list(l1.parameters())

    def run(shared, passes):
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if shared:
            slot = {}
            h0 = decode.linear_act_shared(x, l0, ACT_GELU, slot)
            h1 = decode.linear_act_shared(x, l1, ACT_GELU, slot)
        else:
            h0, h1 = decode.linear_fan(x, [(l0, ACT_GELU, None), (l1, ACT_GELU, None)])
        a, b = (h0 * u0).sum(), (h1 * u1).sum()
        if passes == "a_then_b":
            a.backward(retain_graph=True)
            b.backward()
        else:                                                    # two full passes over the retained graph
            (a + b).backward(retain_graph=True)
            (a + b).backward()
        torch.cuda.synchronize()
        return [x.grad.clone()] + [p.grad.clone() for p in params]

    for passes in ("a_then_b", "twice"):
        ref, got = run(False, passes), run(True, passes)
        for r, g in zip(ref, got):
            _close(g.cpu().numpy(), r.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="shared-input layers, " + passes)


def test_weight_gradients_on_a_side_stream_match_the_inline_ones():
    """decode.overlap_weight_grads(True): sg_weight_grad runs beside the backward chain and joins when backward() returns --
    bit-identical parameter gradients (the kernels are deterministic), also when the pass runs twice in a row."""
    from sings_amd import decode
    dev = _dev()
    torch.manual_seed(3)
    g = decode.GeometryDecoder(n_features=96, isotropic=False).to(dev)
    a = decode.AppearanceDecoder(n_features=96).to(dev)
    x = torch.randn(20000, 96, device=dev)

    def grads():
        for p in list(g.parameters()) + list(a.parameters()):
            p.grad = None
        xi = x.clone().requires_grad_(True)
        g1, a1 = decode.linear_fan(xi, [(g.net[0], decode.ACT_GELU, None), (a.net[0], decode.ACT_GELU, None)])
        og, oa = g(xi, first=g1), a(xi, first=a1)
        loss = sum((v * v).sum() for v in (og['xyz_offsets'], og['rotations'], og['scales'], oa['shs'], oa['opacity']))
        loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in list(g.parameters()) + list(a.parameters())] + [xi.grad.clone()]

    ref = grads()
    decode.overlap_weight_grads(True)
    try:
        for _ in range(2):
            got = grads()
            for r, t in zip(ref, got):
                assert torch.equal(r, t)
    finally:
        decode.overlap_weight_grads(False)


def test_triplane_backward_prepared_early_or_inline():
    """The point-only half of the tri-plane backward is launched by the forward on a side stream by default
    (decode.prepare_triplane_backward_early); switched off, the backward does everything itself: the same gradients
    (plane gradients to rounding -- float atomics --, point gradients bit-identical)."""
    from sings_amd import decode
    dev = _dev()
    torch.manual_seed(11)
    f = _field(dev)
    x0 = (torch.rand(30000, 3, device=dev) * 2.2 - 1.1)
    with torch.no_grad():
        g = torch.randn_like(f(x0))

    def grads():
        for p in f.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        f(x).backward(g)
        torch.cuda.synchronize()
        return x.grad.clone(), [p.grad.clone() for p in f.parameters() if p.requires_grad]

    try:
        decode.prepare_triplane_backward_early(False)
        dx0, dp0 = grads()
        decode.prepare_triplane_backward_early(True)
        dx1, dp1 = grads()
    finally:
        decode._TP["mode"] = None
    assert torch.equal(dx0, dx1)
    for a, b in zip(dp0, dp1):
        _close(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="plane gradient")


@pytest.mark.parametrize("early", [False, True])
@pytest.mark.parametrize("arena", [False, True])
def test_feature_minor_planes_are_used_in_place(early, arena):
    """HexPlaneField(feature_minor=True): the planes keep the reference's shape [1, F, H, W] but live in channels_last memory
    ([H][W][F]); the kernels read them and write their gradients in place (SgTriplane.feature_minor).  Same features bit for bit,
    same point gradients bit for bit, plane gradients to rounding (float atomics) as the plain layout -- with the preparation early
    or inline, with and without a gradient arena; state_dict round trip unchanged; p.grad has the parameter's layout."""
    from sings_amd import decode
    dev = _dev()
    torch.manual_seed(21)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [24, 20, 28], 'multires': [1, 2, 4]}
    f0 = decode.HexPlaneField(cfg, bounds=1.2, device=dev)
    f1 = decode.HexPlaneField(cfg, bounds=1.2, device=dev, feature_minor=True)
    f1.load_state_dict(f0.state_dict())
    for (k0, v0), (k1, v1) in zip(f0.state_dict().items(), f1.state_dict().items()):
        assert k0 == k1 and v0.shape == v1.shape and torch.equal(v0, v1)
    planes1 = [p for p in f1.parameters() if p.requires_grad]
    assert all(p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous() for p in planes1)
    x0 = (torch.rand(40000, 3, device=dev) * 2.6 - 1.3)                       # some points beyond the box: border clamp
    with torch.no_grad():
        y0, y1 = f0(x0), f1(x0)
    assert torch.equal(y0, y1)
    g = torch.randn_like(y0)

    def grads(f):
        params = [p for p in f.parameters() if p.requires_grad]
        for p in params:
            p.grad = None
        views = None
        if arena:
            flat = torch.full((sum(p.numel() for p in params),), float("nan"), device=dev)
            views = decode.set_gradient_arena(params, flat)
        try:
            x = x0.clone().requires_grad_(True)
            f(x).backward(g)
            torch.cuda.synchronize()
            if arena:
                for p, v in zip(params, views):
                    assert p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == p.stride()
            return x.grad.clone(), [p.grad.clone() for p in params]
        finally:
            decode.set_gradient_arena(None, None)

    try:
        decode.prepare_triplane_backward_early(early)
        dx0, dp0 = grads(f0)
        dx1, dp1 = grads(f1)
    finally:
        decode._TP["mode"] = None
    assert torch.equal(dx0, dx1)
    for a, b, p in zip(dp0, dp1, planes1):
        assert b.shape == a.shape and b.stride() == p.stride()
        _close(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="plane gradient (feature-minor)")


def test_gradient_arena_accumulates_like_plain_autograd():
    """with a gradient arena registered, a SECOND backward without a reset in between (micro-batch accumulation,
    zero_grad(set_to_none=False)) wrote the new gradient over p.grad's own memory and autograd then added the slot to itself:
    2 x the last gradient.  The slot is now handed out only while p.grad is None and no earlier hand-out is alive; otherwise
    autograd gets a fresh tensor and adds.  Two backwards over different inputs must give g(x1) + g(x2), bit for bit what
    the same module computes without an arena; after `p.grad = None` the next gradient lands in the slot again."""
    from sings_amd import decode
    dev = _dev()
    torch.manual_seed(5)
    g = decode.GeometryDecoder(n_features=96, isotropic=False).to(dev)
    params = [p for p in g.parameters() if p.requires_grad]
    x1, x2 = torch.randn(6000, 96, device=dev), torch.randn(6000, 96, device=dev)

    def run(xs):
        for x in xs:
            o = g(x)
            sum((v * v).sum() for v in (o['xyz_offsets'], o['rotations'], o['scales'])).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    for p in params:
        p.grad = None
    ref = run([x1, x2])                                     # plain autograd accumulation, no arena
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    views = decode.set_gradient_arena(params, flat)
    try:
        for p in params:
            p.grad = None
        got = run([x1, x2])
        for r, t_ in zip(ref, got):
            assert torch.equal(r, t_)
        # a fresh step: the weight gradients are written in place again
        for p in params:
            p.grad = None
        run([x1])
        in_place = sum(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views))
        assert in_place >= 3, in_place                       # (every nn.Linear weight of the decoder)
    finally:
        decode.set_gradient_arena(None, None)
