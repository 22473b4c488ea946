"""Generates tests/golden/decode_golden.npz by running the REFERENCE's HexPlaneField / GeometryDecoder /
AppearanceDecoder (sings/rec/models/modules/hexplane.py, decoders.py) on CPU in the build container: outputs and
autograd gradients for seeded parameters.  decoders.py imports loguru (absent) -> empty placeholder module.

    python tests/golden/gen_decode_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
lg = types.ModuleType("loguru"); lg.logger = None
sys.modules.setdefault("loguru", lg)
import importlib.util                                               # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


hp = _load("ref_hexplane", "/root/reference/sings/rec/models/modules/hexplane.py")
dc = _load("ref_decoders", "/root/reference/sings/rec/models/modules/decoders.py")

torch.manual_seed(7)
out = {}
cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [16, 12, 10], 'multires': [1, 2]}
field = hp.HexPlaneField(planeconfig=cfg, bounds=1.0, device='cpu')
field.set_aabb([0.9, 1.1, 0.6], [-0.8, -1.0, -0.5])
N = 700
pts = (torch.rand(N, 3) * 2.4 - 1.2)                                 # some points outside the box -> border clamp
pts[:5] = torch.tensor([[0.9, 1.1, 0.6], [-0.8, -1.0, -0.5], [0.05, 0.05, 0.05], [0.9, 0.0, 0.0], [0.0, -1.0, 0.6]])
pts.requires_grad_(True)
feats = field(pts)
w = torch.randn_like(feats)
(feats * w).sum().backward()
out.update(tp_res=np.array(cfg['resolution']), tp_multires=np.array(cfg['multires']), tp_aabb=field.aabb.detach().numpy(),
           tp_pts=pts.detach().numpy(), tp_feats=feats.detach().numpy(), tp_w=w.numpy(), tp_dpts=pts.grad.numpy().copy())
for s, gp in enumerate(field.grids):
    for c, p in enumerate(gp):
        out[f"tp_plane_{s}_{c}"] = p.detach().numpy(); out[f"tp_dplane_{s}_{c}"] = p.grad.numpy().copy()

# decoders on fresh features (n_features = 96 in the shipped config; here 64 = 2 scales x 32)
F_in = 64
x = (torch.randn(N, F_in) * 0.5).requires_grad_(True)
for tag, iso in (("iso", True), ("aniso", False)):
    g = dc.GeometryDecoder(n_features=F_in, isotropic=iso)
    o = g(x)
    loss = (o['xyz_offsets'] * 1.3).sum() + (o['scales'] ** 2).sum() + o['scales_aux'].sum() * 0.1
    if not iso:
        loss = loss + (o['rotations'] * 0.7).sum()
    x.grad = None
    loss.backward()
    out[f"geo_{tag}_x"] = x.detach().numpy()
    for k, v in g.state_dict().items():
        out[f"geo_{tag}_p_{k}"] = v.numpy()
    for k, p in g.named_parameters():
        out[f"geo_{tag}_g_{k}"] = p.grad.numpy().copy()
    for k in ('xyz_offsets', 'scales', 'scales_aux') + (() if iso else ('rotations',)):
        out[f"geo_{tag}_o_{k}"] = o[k].detach().numpy()
    out[f"geo_{tag}_dx"] = x.grad.numpy().copy()
a = dc.AppearanceDecoder(n_features=F_in)
with torch.no_grad():                                              # call-site semantics: sings_hybrid.py:1269-1270
    a.reset_opacity(x.detach())
out["app_offset"] = a.opacity_offset.detach().numpy()
o = a(x)
x.grad = None
((o['shs'] ** 2).sum() * 0.5 + (o['opacity'] * torch.linspace(-1, 1, N)[:, None]).sum()).backward()
for k, v in a.state_dict().items():
    out[f"app_p_{k}"] = v.numpy()
for k, p in a.named_parameters():
    out[f"app_g_{k}"] = p.grad.numpy().copy()
out.update(app_o_shs=o['shs'].detach().numpy(), app_o_opacity=o['opacity'].detach().numpy(), app_dx=x.grad.numpy().copy())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_golden.npz"), **out)
print("wrote decode_golden.npz", len(out), "arrays; feats", feats.shape)
