"""Debug helper (GPU box): run the composed train step with every torch.empty / empty_like allocation pre-filled with a
poison pattern -- any kernel that reads workspace memory before writing it shows up as a changed result.
python tools/poison_empty.py [pattern]   pattern: ff (NaN / -1, default) | 7f | 01"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pat = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0xFF
from sings_amd.decode import AppearanceDecoder, GeometryDecoder, HexPlaneField
from sings_amd.scene import avatar_scene
from sings_amd.train_step import AvatarStep
from sings_amd.regularizers import GaussiansEdgeLoss, L2Norm
from sings_amd.body import joint_transforms
from sings_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device("cuda:0")
N = 150000
s = avatar_scene(N=N, J=52)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
torch.manual_seed(0)
cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [64, 64, 64], 'multires': [1, 2, 4]}
tri = HexPlaneField(cfg, bounds=1.2, device=dev); geo = GeometryDecoder(96).to(dev); app = AppearanceDecoder(96).to(dev)
with torch.no_grad():
    geo.scales[2].bias.fill_(-5.3); geo.scales[2].weight.mul_(0.1); geo.xyz_offsets.weight.mul_(0.01); geo.xyz_offsets.bias.zero_()
cam = s["cam"]
rs = GaussianRasterizationSettings(image_height=s["H"], image_width=s["W"], tanfovx=math.tan(cam["fovx"] * 0.5),
    tanfovy=math.tan(cam["fovy"] * 0.5), bg=t(s["bg"]), scale_modifier=1.0, viewmatrix=t(cam["world_view_transform"]),
    projmatrix=t(cam["full_proj_transform"]), sh_degree=0, campos=t(cam["camera_center"]), prefiltered=False, debug=False)
A = joint_transforms(torch.zeros(3 * s['J'], device=dev), t(s["joints_rest"]), tuple(s["parents"]))
gt = torch.rand(3, s["H"], s["W"], device=dev); ones = torch.ones(s["H"], s["W"], device=dev)
mod = AvatarStep(t(s['xyz_canon']), t(s["lbs_weights"]), tri, geo, app, l2_norm=L2Norm(), gaussian_connect=GaussiansEdgeLoss(),
                 gaussian_connect_w=1.0).to(dev)
params = [p for p in mod.parameters() if p.requires_grad]
smpl_scale, transl, bgt = t(s["smpl_scale"]), t(s["transl"]), t(s["bg"])


def run():
    for p in params: p.grad = None
    loss, ld, ex = mod(A, rs, gt, ones, bgt, smpl_scale=smpl_scale, transl=transl)
    loss.backward()
    torch.cuda.synchronize()
    return {k: float(v) for k, v in ld.items()}, [p.grad.clone() for p in params]


ref, gref = run()
print("clean   ", ref)
_empty, _empty_like = torch.empty, torch.empty_like


def p_empty(*a, **k):
    x = _empty(*a, **k)
    if x.is_cuda and x.numel():
        x.view(torch.uint8).fill_(pat) if x.is_contiguous() else None
    return x


def p_empty_like(*a, **k):
    x = _empty_like(*a, **k)
    if x.is_cuda and x.numel() and x.is_contiguous():
        x.view(torch.uint8).fill_(pat)
    return x


torch.empty, torch.empty_like = p_empty, p_empty_like
try:
    got, ggot = run()
finally:
    torch.empty, torch.empty_like = _empty, _empty_like
print("poisoned", got)
names = [n for n, p in mod.named_parameters() if p.requires_grad]
for n, a, b in zip(names, gref, ggot):
    if not torch.equal(a, b):
        print("   grad differs:", n, float((a - b).abs().max()), "nan" if torch.isnan(b).any() else "")
