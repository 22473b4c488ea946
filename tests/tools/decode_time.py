"""Times the attribute decode on the GPU box: python tests/tools/decode_time.py
ours = HIP tri-plane + bias/activation kernels + library GEMMs; eager = the reference's formulation (the oracle's torch
ops: 9 x F.grid_sample, nn.Linear, GELU) on the same GPU; cpu = the same on the host for a bounded sample."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import decode_oracle as do
from sings_amd.decode import AppearanceDecoder, GeometryDecoder, HexPlaneField, decode_attributes
dev = torch.device("cuda:0")
cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [64, 64, 64], 'multires': [1, 2, 4]}
torch.manual_seed(0)
f = HexPlaneField(cfg, device=dev); g = GeometryDecoder(96).to(dev); a = AppearanceDecoder(96).to(dev)
params = list(f.parameters()) + list(g.parameters()) + list(a.parameters())


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for N in (150000, 500000):
    xyz = (torch.rand(N, 3, device=dev) * 1.8 - 0.9).requires_grad_(True)

    def loss_of(o):
        return o['xyz_canon'].sum() + o['scales'].sum() + o['opacity'].sum() + (o['shs'] ** 2).sum()

    def ours(bwd=True):
        o = decode_attributes(xyz, f, g, a)
        if bwd:
            for p in params: p.grad = None
            xyz.grad = None
            loss_of(o).backward()

    def eager(bwd=True):
        feats = do.triplane_features(xyz, [list(gp) for gp in f.grids], f.aabb)
        og = do.geometry_decoder(feats, dict(g.named_parameters()))
        oa = do.appearance_decoder(feats, dict(a.named_parameters()))
        o = {'xyz_canon': xyz + og['xyz_offsets'], 'scales': og['scales'], 'opacity': oa['opacity'], 'shs': oa['shs']}
        if bwd:
            for p in params: p.grad = None
            xyz.grad = None
            loss_of(o).backward()

    tp_f = timeit(lambda: f(xyz.detach()))
    tp_e = timeit(lambda: do.triplane_features(xyz.detach(), [list(gp) for gp in f.grids], f.aabb))
    print(f"N={N}: tri-plane features fwd  ours {tp_f:8.1f} us | torch eager {tp_e:8.1f} us")
    print(f"N={N}: decode fwd       ours {timeit(lambda: ours(False)):8.1f} us | torch eager {timeit(lambda: eager(False)):8.1f} us")
    print(f"N={N}: decode fwd+bwd   ours {timeit(ours):8.1f} us | torch eager {timeit(eager):8.1f} us")
xs = (torch.rand(20000, 3) * 1.8 - 0.9).requires_grad_(True)
gc = [[p.detach().cpu().requires_grad_(True) for p in gp] for gp in f.grids]
sdg = {k: v.detach().cpu().requires_grad_(True) for k, v in g.named_parameters()}; sda = {k: v.detach().cpu().requires_grad_(True) for k, v in a.named_parameters()}
t0 = time.perf_counter()
feats = do.triplane_features(xs, gc, f.aabb.cpu()); og = do.geometry_decoder(feats, sdg); oa = do.appearance_decoder(feats, sda)
(og['xyz_offsets'].sum() + og['scales'].sum() + oa['opacity'].sum() + (oa['shs'] ** 2).sum()).backward()
print(f"CPU torch, 20 000 points fwd+bwd, {torch.get_num_threads()} threads: {(time.perf_counter() - t0) * 1e3:.1f} ms (x7.5 at 150 k)")
