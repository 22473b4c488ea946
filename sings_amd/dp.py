"""Frame-parallel data parallelism: one process per GPU, RCCL over xGMI.

The reference trains one frame per step on one GPU (gs_trainer.py:207-215, cfg.train.batch_size = 1).
Frames (pose + camera + target image) are independent renders of the SAME canonical Gaussians, so rank
r of W takes frame ``perm[W*t + r]`` at step t (shared-seed permutation), every rank runs the fused
LBS+raster forward/backward on its frame, and ONE sum all-reduce per step combines the canonical-
Gaussian gradients, which the engine already keeps in one flat fp32 buffer (sings_amd/engine.py).
Densification statistics (sings_hybrid.py:1013-1015, gs_trainer.py:486-492) are reduced too (sum, sum,
max) so that every rank takes identical densify / prune decisions.  Forward-only animation / validation
shards frames with no collective at all.

Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
import numpy as np
import torch
import torch.distributed as dist


class FrameSharder:
    """Deterministic frame assignment shared by all ranks (no communication)."""

    def __init__(self, num_frames, world_size, rank, seed=0):
        self.n, self.world, self.rank, self.seed = int(num_frames), int(world_size), int(rank), int(seed)
        if self.n <= 0:
            raise ValueError("num_frames must be positive")

    def _perm(self, epoch):
        return np.random.RandomState(self.seed + 9973 * epoch).permutation(self.n)

    def frame(self, step):
        """Frame index this rank renders at global step ``step``."""
        slot = self.world * int(step) + self.rank
        epoch, pos = divmod(slot, self.n)
        return int(self._perm(epoch)[pos])

    def frames_of_step(self, step):
        return [FrameSharder(self.n, self.world, r, self.seed).frame(step) for r in range(self.world)]

    def eval_frames(self):
        """Forward-only sharding (animation / validation): contiguous strided split, no collective."""
        return list(range(self.rank, self.n, self.world))


class FrameParallel:
    def __init__(self, group=None, average=False, algorithm="all_reduce"):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU, backend 'nccl' = RCCL)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.average = bool(average)
        if algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError("algorithm must be 'all_reduce' or 'rs_ag'")
        self.algorithm = algorithm
        self._pad = None

    def all_reduce_grads(self, flat):
        """Sum (or mean) of the flat canonical-Gaussian gradient buffer over all ranks, in place.

        'rs_ag' = reduce-scatter + all-gather on a world-size-padded copy: on the fully connected xGMI
        mesh (7 links x ~153 GB/s per GPU) each rank exchanges 1/W of the buffer with every peer over a
        different link; 'all_reduce' leaves the schedule to RCCL."""
        if self.world == 1:
            return flat
        if self.algorithm == "all_reduce":
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        else:
            n = flat.numel()
            per = (n + self.world - 1) // self.world
            if self._pad is None or self._pad.numel() != per * self.world or self._pad.device != flat.device:
                self._pad = torch.zeros(per * self.world, dtype=flat.dtype, device=flat.device)
            self._pad[:n].copy_(flat)
            shard = torch.empty(per, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(shard, self._pad, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(self._pad, shard, group=self.group)
            flat.copy_(self._pad[:n])
        if self.average:
            flat.div_(self.world)
        return flat

    def reduce_densification_stats(self, xyz_gradient_accum, denom, max_radii2D):
        """sum / sum / max over ranks, in place (identical topology decisions on every rank)."""
        if self.world == 1:
            return
        dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=self.group)

    def broadcast_(self, tensors, src=0):
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)

    def reduce_scalar(self, value, op="mean"):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / (self.world if op == "mean" else 1)
