"""The per-iteration bookkeeping of ``GaussianTrainer.train`` around ``AvatarStep`` -- what changes BETWEEN steps, and with it the
number of Gaussians every pre-sized piece of the hot path was built for:

    Adam step over the anchors, tri-plane and decoders                      gs_trainer.py:240-262
    densification statistics (screen-space gradient norm, visibility
    count, largest radius) after every backward                             gs_trainer.py:486-492, sings_hybrid.py:1013-1015
    ``oneupSHdegree`` every ``sh_interval`` steps                           gs_trainer.py:436-438
    densify (clone) + prune from the accumulated statistics,
    Adam moments carried over for the survivors                             gs_trainer.py:280-343, sings_hybrid.py:856-932, 968-1004

The reference's ``hybrid`` strategy subdivides the SMPL mesh (trimesh / pytorch3d: its control plane, not built here); this module
follows its ``vanilla`` strategy's clone + prune -- what matters for the hot path is THAT the set changes: P(t) != P(t-1).

A topology change (or a new SH degree) invalidates everything that was sized or captured for the old set:

    the captured HIP graph          ``CapturedStep.replay`` raises until ``AvatarTrainer`` has captured the step again
    the gradient arena              ``decode.arena_sync`` raises for a stale registration; the trainer registers the new parameters
    the pair-capacity hint          re-measured with synchronous steps before the overflow check is deferred again
    Adam's moments, the statistics  re-indexed (kept rows, zeros for the clones) / reset, as the reference does
    the frame-parallel collective   sized by the new flat buffer; the statistics are reduced (sum, sum, max) BEFORE the decision, so every
                                    rank takes the same one (``densify_decision`` is a pure function of the reduced statistics)

Nothing renders with stale sizes: the old anchors' parameter object is dropped, ``AvatarStep.topology_version`` moves on, and every holder
of an old version refuses.
"""
import torch

from . import decode as _dec
from . import rasterizer as _rz
from .train_step import capture_step


def densify_decision(xyz_gradient_accum, denom, scales, opacity, max_grad, scale_threshold, min_opacity, big_scale):
    """-> (clone_mask [N], prune_mask [N + clones]) -- a PURE function of its arguments (every rank that holds the same reduced
    statistics and the same decoded attributes takes the same decision).

    clone   sings_hybrid.py:968-983  ``norm(grads) >= max_grad`` and ``max(scales) <= scale_threshold`` with grads = accum / denom, NaN -> 0
            (:987-988); the clones are appended behind the existing points (``densification_postfix``)
    prune   sings_hybrid.py:1000-1004 on the EXTENDED set: ``opacity < min_opacity`` or ``max(scales) > big_scale`` (``0.1 * extent``).  The
            reference's screen-size term reads ``max_radii2D`` AFTER ``densification_postfix`` has zeroed it (:930-932): it never
            fires, and is left out here."""
    grads = xyz_gradient_accum / denom
    grads = torch.where(torch.isnan(grads), torch.zeros_like(grads), grads)
    smax = scales.reshape(scales.shape[0], -1).max(dim=1).values
    clone = (torch.norm(grads, dim=-1) >= max_grad) & (smax <= scale_threshold)
    op = torch.cat([opacity.reshape(-1), opacity.reshape(-1)[clone]])
    sm = torch.cat([smax, smax[clone]])
    prune = (op < min_opacity) | (sm > big_scale)
    return clone, prune


class CapturedStep:
    """A HIP graph of one training step, valid for ONE topology version of its ``AvatarStep``: ``replay`` refuses (RuntimeError) once
    the set of Gaussians or the SH degree has changed -- the graph holds the old tensors' addresses and the old sizes."""

    def __init__(self, module, fn, warmup=3, device=None):
        self.module, self.version = module, module.topology_version
        self.graph, self.outputs = capture_step(fn, warmup=warmup, device=device)

    def replay(self):
        if self.module.topology_version != self.version:
            raise RuntimeError(f"CapturedStep: captured for topology version {self.version}, the module is at "
                               f"{self.module.topology_version} (densify / prune / SH degree): capture the step again")
        self.graph.replay()
        return self.outputs


class AvatarTrainer:
    """``step(A, gt_rgb, mask)`` = one optimisation step on one posed frame (or a chunk [K,J,4,4]); ``densify_and_prune()`` /
    ``oneup_sh_degree()`` between steps.  ``use_graph``: the step (zero_grad, forward, backward) replays from a HIP graph that is
    captured again after every topology change; Adam and the statistics run outside it.  ``fp``: a ``sings_amd.dp.FrameParallel``
    -- parameter gradients live in one flat arena buffer and are all-reduced, the statistics are reduced before a decision."""

    def __init__(self, module, settings_fn, bg_color, lrs, smpl_scale=None, transl=None, use_graph=False, fp=None,
                 sh_degree=0, max_sh_degree=3, betas=(0.9, 0.999), eps=1e-15):
        self.m, self.settings_fn, self.bg = module, settings_fn, bg_color
        self.smpl_scale, self.transl, self.use_graph, self.fp = smpl_scale, transl, bool(use_graph), fp
        self.sh_degree, self.max_sh_degree = int(sh_degree), int(max_sh_degree)
        self.lrs, self.betas, self.eps = dict(lrs), betas, eps
        self.dev = module.xyz.device
        self.opt = torch.optim.Adam(self._groups(), betas=betas, eps=eps)
        self.captured = self.flat = self.views = None
        self.iteration = 0
        self._static = {}
        self._rebuild()

    # ---- pieces sized by the set of Gaussians
    def _groups(self):
        m = self.m
        return [{"params": [m.xyz], "lr": self.lrs["xyz"], "name": "xyz"},
                {"params": list(m.triplane.parameters()), "lr": self.lrs["triplane"], "name": "triplane"},
                {"params": list(m.geometry_dec.parameters()), "lr": self.lrs["geometry"], "name": "geometry"},
                {"params": list(m.appearance_dec.parameters()), "lr": self.lrs["appearance"], "name": "appearance"}]

    def _rebuild(self):
        """(Re)build what depends on the number of Gaussians / the SH degree: raster settings, statistics, arena, capacity hint, graph."""
        m = self.m
        self.rs = self.settings_fn(self.sh_degree)
        self.params = [p for g in self.opt.param_groups for p in g["params"]]
        m.enable_densification_stats()
        self.captured = None
        if self.fp is not None:
            n = sum(p.numel() for p in self.params)
            self.flat = torch.zeros(n, dtype=torch.float32, device=self.dev)
            self.views = _dec.set_gradient_arena(self.params, self.flat)
        if self.dev.type == "cuda":
            _rz.set_overflow_check("sync")                       # the next steps size the pair capacity for the new set
            _rz.reset_overflow_state()
        self._sizing = 2 if self.use_graph else 0

    def _body(self):
        for p in self.params:
            p.grad = None
        s = self._static
        loss, ld, ex = self.m(s["A"], self.rs, s["gt"], s["mask"], self.bg, smpl_scale=self.smpl_scale, transl=self.transl)
        if loss is None:
            self.m.backward(ld, ex)
        else:
            loss.backward()
        self.m.add_densification_stats(ex)
        return {k: v.detach().reshape(()).clone() for k, v in ld.items()}

    def step(self, A, gt_rgb, mask):
        s = self._static
        if not s or s["A"].shape != A.shape:
            s.update(A=A.clone(), gt=gt_rgb.clone(), mask=mask.clone())
            self.captured = None
        else:
            s["A"].copy_(A); s["gt"].copy_(gt_rgb); s["mask"].copy_(mask)
        if self.use_graph and self._sizing == 0:
            if self.captured is None:
                _rz.set_deferred_overflow_check(True, capacity_pairs=_rz._capacity_hint.get(self.dev.index))
                # (the capture's warm-up runs execute the body -- statistics included -- without being steps: put the statistics back)
                keep = [x.clone() for x in (self.m.xyz_gradient_accum, self.m.denom, self.m.max_radii2D)]
                self.captured = CapturedStep(self.m, self._body, warmup=2, device=self.dev)
                for dst, src in zip((self.m.xyz_gradient_accum, self.m.denom, self.m.max_radii2D), keep):
                    dst.copy_(src)
                _rz.check_deferred_overflow(self.dev)
            ld = self.captured.replay()
        else:
            ld = self._body()                                    # eager (and the synchronous sizing steps in front of a capture)
            self._sizing = max(0, self._sizing - 1)
        if self.fp is not None:
            _dec.arena_sync(self.params, self.views, True)
            self.fp.all_reduce_grads(self.flat)
            _dec.arena_sync(self.params, self.views, False)
        self.opt.step()
        self.iteration += 1
        return ld

    # ---- between steps
    def oneup_sh_degree(self):
        """gs_trainer.py:436-438.  The degree is a field of the raster settings: a captured step must be captured again."""
        if self.sh_degree < self.max_sh_degree:
            self.sh_degree += 1
            self.m.topology_version += 1
            self.rs = self.settings_fn(self.sh_degree)
            self.captured = None

    def densify_and_prune(self, max_grad, scale_threshold, min_opacity, big_scale, decision=None):
        """Clone + prune from the accumulated (and, with ``fp``, reduced) statistics; -> (clone_mask, prune_mask, N_old, N_new).
        ``decision``: a (clone_mask, prune_mask) pair to apply instead (tests drive a reference with the decision of the run)."""
        m = self.m
        if self.fp is not None:
            self.fp.reduce_densification_stats(m.xyz_gradient_accum, m.denom, m.max_radii2D)
        with torch.no_grad():
            if decision is None:
                attrs = _dec.decode_attributes(m.xyz, m.triplane, m.geometry_dec, m.appearance_dec, m.thickness_factor, m.scaling_multiplier)
                decision = densify_decision(m.xyz_gradient_accum, m.denom, attrs["scales"], attrs["opacity"], max_grad, scale_threshold,
                                            min_opacity, big_scale)
            clone, prune = decision
            n_old = int(m.xyz.shape[0])
            keep = ~prune
            # Adam moments: kept rows for the survivors, zeros for the clones (cat_tensors_to_optimizer, _prune_optimizer)
            group = next(g for g in self.opt.param_groups if g["name"] == "xyz")
            old = group["params"][0]
            state = self.opt.state.pop(old, None)
            new_xyz = torch.cat([old.detach(), old.detach()[clone]])[keep].clone()
            new_w = torch.cat([m.lbs_weights, m.lbs_weights[clone]])[keep].clone()
            m.set_topology(new_xyz, new_w)
            group["params"][0] = m.xyz
            if state is not None:
                for k in ("exp_avg", "exp_avg_sq"):
                    state[k] = torch.cat([state[k], torch.zeros_like(state[k][clone])])[keep].clone()
                self.opt.state[m.xyz] = state
        self._rebuild()
        return clone, prune, n_old, int(m.xyz.shape[0])
