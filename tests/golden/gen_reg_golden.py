"""Generates tests/golden/reg_golden.npz by running the REFERENCE's regulariser classes
(sings/rec/losses/loss_items.py: L2Norm, GaussiansEdgeLoss, RegionLaplacianLoss_v2 incl. forward_hands) in the build
container on seeded inputs, values and autograd gradients.

    python tests/golden/gen_reg_golden.py

The module imports pytorch3d (knn_points, laplacian, cot_laplacian, norm_laplacian) and smpl_parsing.parse_weights (which
loads licensed SMPL region json files) at the top; neither is available here.  Placeholder modules provide the two
primitives actually used, restated from pytorch3d's published definitions (oracle/reg_oracle.py), and a parse_weights
that orders the weight dict by its keys' position (label k = k-th region of human_complex.yaml:137-142; hands = 6, 7).
L2Norm is pure torch -> fully the reference's arithmetic.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from oracle import reg_oracle as ro                               # noqa: E402

p3 = types.ModuleType("pytorch3d"); ops = types.ModuleType("pytorch3d.ops")
ops.knn_points = ro.knn_points; ops.laplacian = ro.laplacian; ops.cot_laplacian = object(); ops.norm_laplacian = object()
sys.modules["pytorch3d"] = p3; sys.modules["pytorch3d.ops"] = ops
sp = types.ModuleType("sings.rec.utils.body_model.smpl_parsing")
sp.parse_weights = lambda d: np.array(list(d.values()), dtype=np.float64)
sys.modules["sings.rec.utils.body_model.smpl_parsing"] = sp
from sings.rec.losses import loss_items as ref                    # noqa: E402

REGIONS = ['head-neck', 'spine', 'leftUpArm', 'rightUpArm', 'leftDownArm', 'rightDownArm', 'leftHand', 'rightHand', 'hips',
           'leftUpLeg', 'rightUpLeg', 'leftDownLeg', 'rightDownLeg', 'leftFoot', 'rightFoot']
POS_W = dict(zip(REGIONS, [0.5, 0.75, 1., 1., 1., 1., 1.5, 1.5, 1., 1., 1., 1., 1., 0.75, 0.75]))     # human_complex.yaml:141
COL_W = dict(zip(REGIONS, [0., 0., 0., 0., 1., 1., 1., 1., 0., 0., 0., 0., 0., 0., 0.]))               # :137

rs = np.random.RandomState(5)
out = {}
# ---- mesh: a 30 x 45 triangulated sheet, 15 label stripes of 3 columns (every vertex has same-label edges)
nx, ny = 30, 45
gx, gy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
verts = np.stack([gx.ravel() * 0.02, gy.ravel() * 0.02, 0.05 * np.sin(gx.ravel() * 0.4) * np.cos(gy.ravel() * 0.3)], 1).astype(np.float32)
verts += rs.normal(0, 0.003, verts.shape).astype(np.float32)
vid = lambda i, j: i * ny + j
E = set()
for i in range(nx):
    for j in range(ny):
        if i + 1 < nx: E.add((vid(i, j), vid(i + 1, j)))
        if j + 1 < ny: E.add((vid(i, j), vid(i, j + 1)))
        if i + 1 < nx and j + 1 < ny: E.add((vid(i, j), vid(i + 1, j + 1)))
edges = np.array(sorted(E), dtype=np.int64)
labels = (gy.ravel() // 3).astype(np.int64)
out.update(mesh_verts=verts, mesh_edges=edges, mesh_labels=labels)
tv, te, tl = torch.from_numpy(verts), torch.from_numpy(edges), torch.from_numpy(labels)
for tag, W, C in (("pos", POS_W, 3), ("col", COL_W, 3)):
    mod = ref.RegionLaplacianLoss_v2(verts=tv, edges=te, vertex_labels=tl, region_weights=W)
    x = (tv + 0.01 * torch.randn(tv.shape, generator=torch.Generator().manual_seed(3 if tag == "pos" else 4))).requires_grad_(True)
    loss = mod(x); loss.backward()
    out.update({f"lap_{tag}_x": x.detach().numpy(), f"lap_{tag}_loss": np.float32(loss.item()), f"lap_{tag}_grad": x.grad.numpy().copy(),
                f"lap_{tag}_w": np.array(list(W.values()), np.float32)})
    if tag == "pos":
        x2 = x.detach().clone().requires_grad_(True)
        lh = mod.forward_hands(x2); lh.backward()
        out.update(lap_hands_loss=np.float32(lh.item()), lap_hands_grad=x2.grad.numpy().copy())
# ---- Gaussians: points on a wavy surface + clusters, isotropic-ish scales
N = 5000
u, v = rs.uniform(0, 1, N), rs.uniform(0, 1.8, N)
xyz = np.stack([u, v, 0.1 * np.sin(6 * u) * np.cos(4 * v)], 1).astype(np.float32)
xyz[:500] = (xyz[:500] * 0.05 + np.array([0.3, 0.9, 0.0])).astype(np.float32)          # a dense blob (hands-like)
xyz[500:520] = xyz[520:540]                                                             # exact duplicates
scales = np.exp(rs.normal(-4.6, 0.5, (N, 3))).astype(np.float32)
opacity = rs.uniform(0.01, 0.99, (N, 1)).astype(np.float32)
offsets = rs.normal(0, 0.004, (N, 3)).astype(np.float32)
out.update(gs_xyz=xyz, gs_scales=scales, gs_opacity=opacity, gs_offsets=offsets)
T = lambda a: torch.from_numpy(a)
sc = T(scales).requires_grad_(True)
loss = ref.GaussiansEdgeLoss()({'xyz_canon': T(xyz), 'scales': sc}); loss.backward()
out.update(edge_loss=np.float32(loss.item()), edge_grad_scales=sc.grad.numpy().copy())
lam = dict(lambda_xyz_offsets=0.001, lambda_scales_diff=0.005, max_scale_threshold=0.005, lambda_max_scale=0.01,
           min_opacity_threshold=0.2, lambda_min_opacity=0.001)                           # human_complex.yaml:148-154
for tag, keys in (("full", ("xyz_offsets", "scales", "opacity")), ("noop", ("xyz_offsets", "scales"))):
    ins = {'xyz_offsets': T(offsets).requires_grad_(True), 'scales': T(scales).requires_grad_(True), 'opacity': T(opacity).requires_grad_(True)}
    d = {k: ins[k] for k in keys}
    l = ref.L2Norm(**lam)(d); l.backward()
    out[f"l2_{tag}_loss"] = np.float32(l.item())
    for k in keys:
        out[f"l2_{tag}_grad_{k}"] = ins[k].grad.numpy().copy()
out["l2_lambdas"] = np.array([lam["lambda_xyz_offsets"], lam["lambda_scales_diff"], lam["lambda_max_scale"], lam["max_scale_threshold"],
                              lam["lambda_min_opacity"], lam["min_opacity_threshold"]], np.float32)
# mesh_edge_loss: pytorch3d.loss is not importable -> definition only (oracle), recorded for the GPU test
mv = tv.clone().requires_grad_(True)
ml = ro.mesh_edge_loss(mv, te); ml.backward()
out.update(mesh_edge_loss=np.float32(ml.item()), mesh_edge_grad=mv.grad.numpy().copy())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reg_golden.npz"), **out)
print("wrote reg_golden.npz", {k: (v.shape if hasattr(v, 'shape') and v.shape else float(v)) for k, v in out.items() if 'loss' in k})
