"""A training LOOP on the GPU across a topology change: 40 Adam steps through ``sings_amd.train_loop.AvatarTrainer``
(eager, and replayed from a HIP graph that is captured again after the change) against the same optimiser driven on the CPU by the
gradients of the oracle chain (decode_oracle -> lbs_oracle -> raster oracle (C) -> photo_loss_oracle) -- the loss TRAJECTORY, not one step.
At step 20 the set of Gaussians is densified (clones) and pruned from the accumulated statistics and the SH degree is raised: the number
of Gaussians changes under the captured graph, the gradient arena and the pair-capacity hint, and every holder of the old set must
refuse instead of rendering with stale sizes.

What the reference does between steps: gs_trainer.py:240-262 (Adam), :436-438 (oneupSHdegree), :280-343, 486-521 (statistics, densify /
prune), sings_hybrid.py:856-932, 968-1004, 1013-1015.

Tolerance of the trajectory: both sides compute in fp32 with different summation orders (the tri-plane backward on the GPU adds with
float atomics), and Adam divides by sqrt(v) + 1e-15: a parameter whose gradient is noise moves by a full learning rate in a direction
that rounding decides.  Bound: 1e-4 relative at every step (measured on the MI355X: 7e-7 against the oracle-driven optimiser, 2.5e-7 between
the eager and the graph-replayed loop)."""
import math

import numpy as np
import pytest
import torch

from oracle import decode_oracle as do
from oracle import lbs_oracle as lo
from oracle import photo_loss_oracle as plo
from oracle import raster_oracle as ro
from sings_amd.camera import make_camera

pytestmark = pytest.mark.gpu

N0, J, W, H, STEPS, CHANGE_AT = 4000, 24, 128, 224, 40, 20
LRS = {"xyz": 2e-4, "triplane": 2e-3, "geometry": 5e-4, "appearance": 1e-3}


def _scene():
    rs = np.random.RandomState(7)
    xyz = (rs.normal(0, 0.3, (N0, 3)) * np.array([0.45, 0.9, 0.3])).astype(np.float32)
    w = rs.rand(N0, J).astype(np.float32) ** 6
    w[np.arange(N0), rs.randint(0, J, N0)] += 0.3
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    frames = []
    for _ in range(8):                                                            # eight poses, visited round-robin
        A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1))
        for j in range(J):
            A[j, :3, :3] = lo.batch_rodrigues(torch.from_numpy(rs.normal(0, 0.2, (1, 3)).astype(np.float32))).numpy()[0]
            A[j, :3, 3] = rs.normal(0, 0.03, 3)
        frames.append(A)
    cam = make_camera(np.eye(4, dtype=np.float32), 700.0, 700.0, W / 2, H / 2, W, H)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((((xx - W / 2) / (W / 2.6)) ** 2 + ((yy - H / 2) / (H / 2.3)) ** 2) < 1).astype(np.float32)
    gts = [rs.uniform(0, 1, (3, H, W)).astype(np.float32) for _ in range(8)]
    return dict(xyz=xyz, w=w, frames=frames, cam=cam, mask=mask, gts=gts, transl=np.array([0.01, -0.03, 4.2], np.float32),
                smpl_scale=np.array([1.03], np.float32), bg=np.array([0.3, 0.5, 0.2], np.float32))


def _modules(dev, seed=11):
    from sings_amd.decode import AppearanceDecoder, GeometryDecoder, HexPlaneField
    torch.manual_seed(seed)
    cfg = {'grid_dimensions': 2, 'input_coordinate_dim': 3, 'output_coordinate_dim': 32, 'resolution': [16, 16, 16], 'multires': [1, 2]}
    tri = HexPlaneField(cfg, device=dev); geo = GeometryDecoder(64).to(dev); app = AppearanceDecoder(64).to(dev)
    with torch.no_grad():
        geo.scales[2].bias.fill_(-3.7); geo.xyz_offsets.weight.mul_(0.05); geo.xyz_offsets.bias.mul_(0.05)
    return tri, geo, app


class _CpuReference:
    """The same model on the CPU: parameters copied from the GPU modules, gradients from the oracle chain, torch.optim.Adam."""

    def __init__(self, sc, tri, geo, app):
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        self.sc, self.T = sc, T
        self.xyz = T(sc["xyz"]).clone().requires_grad_(True)
        self.w = T(sc["w"]).clone()
        self.grids = [[p.detach().cpu().clone().requires_grad_(True) for p in gp] for gp in tri.grids]
        self.aabb = tri.aabb.detach().cpu()
        self.sdg = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in geo.named_parameters()}
        self.sda = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in app.named_parameters()}
        self.opt = torch.optim.Adam([{"params": [self.xyz], "lr": LRS["xyz"]},
                                     {"params": [p for gp in self.grids for p in gp], "lr": LRS["triplane"]},
                                     {"params": list(self.sdg.values()), "lr": LRS["geometry"]},
                                     {"params": list(self.sda.values()), "lr": LRS["appearance"]}], betas=(0.9, 0.999), eps=1e-15)
        n = self.xyz.shape[0]
        self.accum, self.denom, self.maxr = np.zeros((n, 1), np.float32), np.zeros((n, 1), np.float32), np.zeros(n, np.float32)
        self.sh_degree = 0

    def step(self, f):
        sc, T = self.sc, self.T
        cam = sc["cam"]
        self.opt.zero_grad(set_to_none=True)
        n = self.xyz.shape[0]
        feats = do.triplane_features(self.xyz, self.grids, self.aabb)
        og = do.geometry_decoder(feats, self.sdg); oa = do.appearance_decoder(feats, self.sda)
        posed = lo.deform_gaussians(self.xyz + og['xyz_offsets'], torch.eye(3)[None].repeat(n, 1, 1), og['scales'], self.w, T(sc["frames"][f]),
                                    smpl_scale=T(sc["smpl_scale"]), transl=T(sc["transl"]))
        pxyz, prot, psc, _ = posed
        o = ro.forward(pxyz.detach().numpy(), oa['opacity'].detach().numpy(), cam["world_view_transform"], cam["full_proj_transform"],
                       cam["camera_center"], W, H, math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), sc["bg"],
                       scales=psc.detach().numpy(), rotations=prot.detach().numpy(), shs=oa['shs'].detach().numpy(), sh_degree=self.sh_degree)
        img = T(o["color"]).requires_grad_(True)
        pl = plo.photometric_loss(img, T(sc["gts"][f]), T(sc["mask"]), T(sc["bg"]), 0.8, 0.2)
        loss = pl["l1"] + pl["ssim"]
        loss.backward()
        g = ro.backward(o, img.grad.numpy())
        pairs = [(pxyz, T(g["dL_dmeans3D"])), (psc, T(g["dL_dscales"])), (prot, T(g["dL_drots"])),
                 (oa['opacity'], T(g["dL_dopacity"]).reshape(n, 1)), (oa['shs'], T(g["dL_dsh"]).reshape(n, 16, 3))]
        pairs = [(a, b) for a, b in pairs if a.requires_grad]            # (identity canonical rotations under fixed joint transforms: no graph)
        torch.autograd.backward([a for a, _ in pairs], [b for _, b in pairs])
        vis = o["radii"] > 0
        self.accum[vis, 0] += np.linalg.norm(g["dL_dmean2D"][vis, :2], axis=-1)
        self.denom[vis] += 1
        self.maxr[vis] = np.maximum(self.maxr[vis], o["radii"][vis])
        self.opt.step()
        self.last = dict(scales=og['scales'].detach(), opacity=oa['opacity'].detach(), R=o["R"])
        return float(loss.detach())

    def apply(self, clone, prune):
        """The run's decision on the CPU model: clones appended, pruned rows dropped, Adam moments kept / zeroed, statistics reset."""
        clone, keep = clone.cpu(), ~prune.cpu()
        old = self.xyz
        st = self.opt.state.pop(old)
        self.xyz = torch.cat([old.detach(), old.detach()[clone]])[keep].clone().requires_grad_(True)
        self.w = torch.cat([self.w, self.w[clone]])[keep].clone()
        for k in ("exp_avg", "exp_avg_sq"):
            st[k] = torch.cat([st[k], torch.zeros_like(st[k][clone])])[keep].clone()
        self.opt.param_groups[0]["params"][0] = self.xyz
        self.opt.state[self.xyz] = st
        n = self.xyz.shape[0]
        self.accum, self.denom, self.maxr = np.zeros((n, 1), np.float32), np.zeros((n, 1), np.float32), np.zeros(n, np.float32)


def _trainer(sc, dev, use_graph):
    from sings_amd.rasterizer import GaussianRasterizationSettings
    from sings_amd.train_loop import AvatarTrainer
    from sings_amd.train_step import AvatarStep
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tri, geo, app = _modules(dev)
    mod = AvatarStep(t(sc["xyz"]), t(sc["w"]), tri, geo, app).to(dev)
    cam = sc["cam"]
    consts = dict(bg=t(sc["bg"]), view=t(cam["world_view_transform"]), proj=t(cam["full_proj_transform"]), campos=t(cam["camera_center"]))

    def settings(deg):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5), tanfovy=math.tan(cam["fovy"] * 0.5), bg=consts["bg"],
            scale_modifier=1.0, viewmatrix=consts["view"], projmatrix=consts["proj"], sh_degree=deg, campos=consts["campos"],
            prefiltered=False, debug=False)
    tr = AvatarTrainer(mod, settings, consts["bg"], LRS, smpl_scale=t(sc["smpl_scale"]), transl=t(sc["transl"]), use_graph=use_graph,
                       sh_degree=0, max_sh_degree=1)
    return tr, (tri, geo, app)


def _pick_thresholds(accum, denom, opacity):
    """Thresholds that clone ~10 % and prune ~5 % of THIS run's Gaussians (the reference's are tuned to its real data)."""
    g = (accum / denom.clamp_min(1)).reshape(-1)
    return float(torch.quantile(g, 0.90)), float(torch.quantile(opacity.reshape(-1), 0.05))


def test_forty_steps_across_a_densify_prune_and_an_sh_degree_change():
    from sings_amd import decode, rasterizer as rz
    from sings_amd.train_loop import CapturedStep, densify_decision
    dev = torch.device("cuda:0")
    sc = _scene()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eager, mods = _trainer(sc, dev, use_graph=False)
    graph, _ = _trainer(sc, dev, use_graph=True)
    ref = _CpuReference(sc, *mods)
    A = [t(a) for a in sc["frames"]]; gts = [t(g) for g in sc["gts"]]; mask = t(sc["mask"])
    losses = {"eager": [], "graph": [], "cpu": []}
    stale = None
    try:
        for it in range(STEPS):
            f = it % 8
            if it == CHANGE_AT:
                # ---- statistics of the first 20 steps: the GPU's accumulation against the oracle's
                m = eager.m
                acc, den, mr = m.xyz_gradient_accum.cpu().numpy(), m.denom.cpu().numpy(), m.max_radii2D.cpu().numpy()
                assert np.array_equal(den, ref.denom) and den.max() == CHANGE_AT, "visibility counts"
                assert np.abs(acc - ref.accum).max() <= 2e-3 * np.abs(ref.accum).max()
                assert (mr == ref.maxr).mean() > 0.999
                with torch.no_grad():
                    attrs = decode.decode_attributes(m.xyz, m.triplane, m.geometry_dec, m.appearance_dec, m.thickness_factor, m.scaling_multiplier)
                max_grad, min_op = _pick_thresholds(m.xyz_gradient_accum, m.denom, attrs["opacity"])
                dec = densify_decision(m.xyz_gradient_accum, m.denom, attrs["scales"], attrs["opacity"], max_grad, 1.0, min_op, 1e9)
                # the oracle-driven model would decide the same for all but borderline Gaussians
                dref = densify_decision(torch.from_numpy(ref.accum), torch.from_numpy(ref.denom), ref.last["scales"], ref.last["opacity"],
                                        max_grad, 1.0, min_op, 1e9)
                n_old = int(m.xyz.shape[0])
                assert (dec[0].cpu() != dref[0]).sum() <= 0.002 * n_old
                # ---- the topology change, the SAME decision everywhere; the captured graph of the old set is kept to be refused
                stale = graph.captured
                assert isinstance(stale, CapturedStep) and stale.version == graph.m.topology_version
                clone, prune, n0, n1 = eager.densify_and_prune(max_grad, 1.0, min_op, 1e9, decision=dec)
                graph.densify_and_prune(max_grad, 1.0, min_op, 1e9, decision=dec)
                ref.apply(clone, prune)
                n_clone, n_prune = int(clone.sum()), int(prune.sum())
                assert 0.05 * n0 <= n_clone <= 0.15 * n0 and 0.02 * n0 <= n_prune <= 0.1 * n0 and n1 == n0 + n_clone - n_prune != n0
                assert int(eager.m.xyz.shape[0]) == int(graph.m.xyz.shape[0]) == int(ref.xyz.shape[0]) == n1
                assert eager.m.lbs_weights.shape[0] == n1 and eager.m.denom.shape[0] == n1 and float(eager.m.denom.abs().max()) == 0.0
                for tr_ in (eager, graph):
                    tr_.oneup_sh_degree()
                ref.sh_degree = 1
                assert eager.sh_degree == graph.sh_degree == 1 and eager.rs.sh_degree == 1
                # the old graph refuses; the trainer has dropped it and captures again
                with pytest.raises(RuntimeError, match="capture the step again"):
                    stale.replay()
                assert graph.captured is None
            le = eager.step(A[f], gts[f], mask)
            lg = graph.step(A[f], gts[f], mask)
            lc = ref.step(f)
            losses["eager"].append(float(le["loss"])); losses["graph"].append(float(lg["loss"])); losses["cpu"].append(lc)
            assert int(eager.m.xyz.grad.shape[0]) == int(eager.m.xyz.shape[0])
        rz.check_deferred_overflow(dev)                             # nothing overflowed inside the captured steps
    finally:
        rz.set_overflow_check("sync"); rz.reset_overflow_state()
        decode.set_gradient_arena(None, None)
    e, g, c = (np.array(losses[k]) for k in ("eager", "graph", "cpu"))
    rel_ec, rel_eg = np.abs(e - c) / np.abs(c), np.abs(e - g) / np.abs(e)
    print(f"loss trajectory: first {c[0]:.6f} last {c[-1]:.6f}; max rel eager-vs-oracle {rel_ec.max():.2e} (step {rel_ec.argmax()}), "
          f"eager-vs-graph {rel_eg.max():.2e}; N {N0} -> {int(eager.m.xyz.shape[0])}")
    assert c[-1] < c[0], "the loop optimises"
    assert rel_ec.max() <= 1e-4, (rel_ec.argmax(), rel_ec.max())
    assert rel_eg.max() <= 1e-4, (rel_eg.argmax(), rel_eg.max())
    assert graph.captured is not None and graph.captured.version == graph.m.topology_version      # captured again for the new set


def test_stale_holders_refuse_after_a_topology_change():
    """The gradient arena registered for the old parameters and an arena_sync with the old parameter list raise; a fresh registration
    works; a chunk of frames [K,J,4,4] keeps one statistic per Gaussian (sum over its frames)."""
    from sings_amd import decode
    dev = torch.device("cuda:0")
    sc = _scene()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tr, _ = _trainer(sc, dev, use_graph=False)
    try:
        params = list(tr.params)
        flat = torch.zeros(sum(p.numel() for p in params), device=dev)
        views = decode.set_gradient_arena(params, flat)
        A2 = torch.stack([t(sc["frames"][0]), t(sc["frames"][1])])
        gt2 = torch.stack([t(sc["gts"][0]), t(sc["gts"][1])])
        tr.step(A2, gt2, t(sc["mask"]))                                            # K = 2 frames in one step
        decode.arena_sync(params, views, True)
        den = tr.m.denom
        assert float(den.max()) == 2.0 and tr.m.xyz_gradient_accum.shape == (N0, 1)
        keep = torch.ones(N0, dtype=torch.bool, device=dev); keep[::7] = False
        tr.m.set_topology(tr.m.xyz.detach()[keep], tr.m.lbs_weights[keep])
        with pytest.raises(RuntimeError, match="stale"):
            decode.arena_sync(params, views, True)
        new_params = [p for p in tr.m.parameters() if p.requires_grad]
        with pytest.raises(RuntimeError, match="stale"):
            decode.arena_sync(new_params, views, True)
        flat2 = torch.zeros(sum(p.numel() for p in new_params), device=dev)
        views2 = decode.set_gradient_arena(new_params, flat2)
        with pytest.raises(RuntimeError, match="shapes"):
            decode.arena_sync(new_params, [views[0]] + views2[1:], True)                # (the old anchors' view: [N0,3] against [N1,3])
        assert decode.arena_sync(new_params, views2, True) >= 0
    finally:
        decode.set_gradient_arena(None, None)
