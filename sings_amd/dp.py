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
    def __init__(self, group=None, average=False, algorithm="all_reduce", host_staged=False, force=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU, backend 'nccl' = RCCL)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.average = bool(average)
        if algorithm not in ("all_reduce", "rs_ag"):
            raise ValueError("algorithm must be 'all_reduce' or 'rs_ag'")
        self.algorithm = algorithm
        # host_staged: device tensors take their collectives through a host copy (synchronous).  Only for process groups
        # whose backend cannot see device memory -- gloo, used when several ranks have to share ONE GPU (single-GPU test
        # boxes: RCCL refuses two ranks on a device).  Never the multi-GPU path.
        self.host_staged = bool(host_staged)
        # force: issue every collective even in a ONE-rank group (RCCL accepts a one-rank communicator; the result equals the
        # input bit for bit).  This is how the nccl branches below -- async all-reduce, in-place reduce-scatter + all-gather,
        # work handles, chunked pipeline -- are executed on a single-GPU box (tests/test_gpu_bench.py, bench.py under
        # SINGS_BENCH_FORCE_DIST=1); without it a world of one returns before touching the backend.
        self.force = bool(force)

    @property
    def active(self):
        """True when all_reduce_grads really issues collectives (several ranks, or a forced one-rank group)."""
        return self.world > 1 or self.force

    def all_reduce_grads(self, flat, async_op=False):
        """Sum (or mean) of the flat canonical-Gaussian gradient buffer over all ranks, in place.

        'rs_ag' = reduce-scatter + all-gather: on the fully connected xGMI mesh (7 links x ~153 GB/s per
        GPU) each rank exchanges 1/W of the buffer with every peer over a different link; 'all_reduce'
        leaves the schedule to RCCL.  The collective runs in place on ``flat``; when its length is not a
        multiple of the world size only the ragged tail (< W elements) takes a second, tiny all-reduce --
        the bulk is never copied.  ``async_op=True`` (device tensors): the collectives are queued on the
        backend's stream behind the work of the CURRENT stream and a list of work handles is returned;
        ``FrameParallel.wait(handles, flat)`` makes the current stream wait for them (and applies the
        mean)."""
        if not self.active:
            return [] if async_op else flat
        if self.host_staged and flat.is_cuda:
            h = flat.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            if self.average and not async_op:
                h.div_(self.world)
            flat.copy_(h)
            return [] if async_op else flat
        works = []
        if self.algorithm == "all_reduce":
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        else:
            n, w = flat.numel(), self.world
            main = n // w * w
            if main:
                body = flat[:main]
                shard = body.view(w, main // w)[self.rank]
                works.append(dist.reduce_scatter_tensor(shard, body, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
                works.append(dist.all_gather_into_tensor(body, shard, group=self.group, async_op=async_op))
            if n - main:
                works.append(dist.all_reduce(flat[main:], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        if async_op:
            return works
        if self.average:
            flat.div_(self.world)
        return flat

    def wait(self, works, flat=None):
        for w in works:
            w.wait()
        if flat is not None and self.average and self.world > 1:
            flat.div_(self.world)

    def reduce_densification_stats(self, xyz_gradient_accum, denom, max_radii2D):
        """sum / sum / max over ranks, in place (identical topology decisions on every rank)."""
        if not self.active:
            return
        for t, op in ((xyz_gradient_accum, dist.ReduceOp.SUM), (denom, dist.ReduceOp.SUM), (max_radii2D, dist.ReduceOp.MAX)):
            if self.host_staged and t.is_cuda:                    # (gloo group over device tensors: single-GPU test boxes only)
                h = t.detach().cpu()
                dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op, group=self.group)

    def broadcast_(self, tensors, src=0):
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)

    def reduce_scalar(self, value, op="mean"):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / (self.world if op == "mean" else 1)


class GradientPipeline:
    """Gradient rows of the K views of one step -> ONE summed (and, with W > 1 ranks, all-reduced) buffer
    (SURVEY.md 8(e)).  Runs on the CALLER's stream, after the views have been joined into it:

        for every chunk c:   acc[c] = rows[0][c] + ... + rows[K-1][c]      (one pass over the chunk, fixed order)
                             all-reduce(acc[c]) queued asynchronously      (RCCL's own stream, behind this fold)
        wait for the collectives

    so the fold of chunk c+1 runs under the transfer of chunk c, and with one rank this is the plain one-pass sum.
    What no schedule can hide is one full-size all-reduce after the last view: every gradient element depends on the last
    backward kernel of the last view, and the optimiser needs the reduced sum before the next forward.

    Two schedules that fold rows EARLIER, on a dedicated communication stream fed by per-view events, were built and
    measured on the MI355X and are not in the tree: folding every view on arrival 2 917 views/s, folding all but the views
    still in flight 3 060, against 3 278 for this one (cfg3, 8 views, 3 streams, one GPU); the avatar step lost 27 %.
    Two causes: the extra 141-MB read-modify-write passes compete with the latency-bound binning kernels of the views
    still rendering, and -- larger -- a FIFTH stream next to the caller's and the three rendering streams: HIP multiplexes
    streams onto four hardware queues, so the communication stream shares a queue with a rendering stream and its
    event waits stall that stream's kernels.  Works on CPU tensors too (gloo tests)."""

    def __init__(self, rows, frame_parallel=None, chunks=4, active=None):
        """``active``: only ``rows[:, :active]`` carries gradient (engines with ``sh_planar``: the SH planes not in use stay zero
        behind it) -- only that prefix is folded and all-reduced; ``acc[active:]`` is zero."""
        if rows.dim() != 2:
            raise ValueError("rows must be [views, floats_per_view]")
        self.rows, self.fp = rows, frame_parallel
        self.k, self.n_full = rows.shape
        self.n = self.n_full if active is None else int(active)
        if not 0 < self.n <= self.n_full:
            raise ValueError("active must be in (0, floats_per_view]")
        self.cuda = rows.is_cuda
        # one view per step: the row IS the sum (no fold, no copy)
        self.acc = rows[0] if self.k == 1 else torch.zeros(self.n_full, dtype=rows.dtype, device=rows.device)
        self._chunks = chunks
        self._set_bounds()
        if self.cuda:
            self._t = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self._timed = False

    def _set_bounds(self):
        fp = self.fp
        world = 1 if fp is None else fp.world
        c = max(1, int(self._chunks)) if fp is not None and fp.active else 1
        step = -(-self.n // c)
        step = -(-step // (world * 64)) * (world * 64)           # chunk boundaries the rs_ag schedule divides, 256-B aligned
        self.bounds = [(lo, min(lo + step, self.n)) for lo in range(0, self.n, step)]

    def set_active(self, active):
        """Change the prefix that is folded and reduced (``oneupSHdegree``, gs_trainer.py:436-438: more SH planes carry gradient
        from now on).  Growing is always safe -- the planes behind the old prefix were zero; every rank must call it at the same
        step (the chunk boundaries of the collective follow)."""
        active = self.n_full if active is None else int(active)
        if not 0 < active <= self.n_full:
            raise ValueError("active must be in (0, floats_per_view]")
        if active < self.n:
            self.acc[active:self.n].zero_()                      # (what the wider fold -- k == 1: the row itself -- left there)
        self.n = active
        self._set_bounds()

    def reduce(self):
        """Call on the stream the views were joined into.  Returns ``acc`` (ready on that stream)."""
        works = []
        if self.cuda and self._timed:
            self._t[0].record(torch.cuda.current_stream(self.rows.device))      # = the last view's gradients exist
        for lo, hi in self.bounds:
            a = self.acc[lo:hi]
            if self.k > 1:
                torch.sum(self.rows[:, lo:hi], dim=0, out=a)
            if self.fp is not None:
                if self.cuda:
                    works += self.fp.all_reduce_grads(a, async_op=True)        # the next chunk's fold does not wait for it
                else:
                    self.fp.all_reduce_grads(a)
        if self.fp is not None and self.cuda:
            self.fp.wait(works, None)
            if self.fp.average and self.fp.world > 1:
                self.acc[:self.n].div_(self.fp.world)
        if self.cuda and self._timed:
            self._t[1].record(torch.cuda.current_stream(self.rows.device))
        return self.acc

    def one_shot(self):
        """Reference schedule: fold everything, then ONE collective over the whole buffer (the pre-round-2 behaviour of
        bench.py).  ``reduce`` gives the same SUM; bit-identical to this only where the backend's reduction order does not
        depend on an element's position in the buffer -- true for two ranks (one addition per element: what
        tests/test_dp_gloo.py checks) and for a forced one-rank group, NOT in general for RCCL ring / tree schedules with more
        ranks, where chunking changes which rank adds first: there both schedules are deterministic per configuration, and
        equal only to rounding."""
        if self.k > 1:
            torch.sum(self.rows[:, :self.n], dim=0, out=self.acc[:self.n])
        if self.fp is not None:
            self.fp.all_reduce_grads(self.acc[:self.n])
        return self.acc

    # -- measurement --------------------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._timed = bool(on) and self.cuda

    def exposed_ms(self):
        """After a synchronised step run with timing enabled: time from "last view's gradients exist" to "reduced sum
        ready" = the part of the reduction nothing can hide."""
        return self._t[0].elapsed_time(self._t[1])
