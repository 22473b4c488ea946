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
    def __init__(self, group=None, average=False, algorithm="all_reduce", host_staged=False):
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
        if self.world == 1:
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


class GradientPipeline:
    """Gradient rows of the K views of one step -> ONE summed (and, with W > 1 ranks, all-reduced) buffer, off the
    critical path of the rendering streams ("overlapped with the tail of backward-preprocess").

    The views of a step finish one after another (sings_amd.engine.ViewBatch deals them to a few HIP streams).  As soon as
    view v's backward has been queued, ``view_done(v)`` records an event on ITS stream; a dedicated communication stream
    waits for those events and folds the rows into ``acc`` in a FIXED order (so the sum is bit-identical however the views
    were interleaved, and identical to ``one_shot``):

        acc  = rows[0] + ... + rows[h-1]     one pass, as soon as the first h = K - tail views are done (hidden under the
                                             `tail` views that are still rendering: one view per rendering stream)
        acc += rows[v]   for v = h .. K-2    as each of them finishes (hidden under the views after it)
        acc += rows[K-1]                     chunk by chunk; every chunk goes into its collective the moment it is
                                             complete, so the fold of chunk c+1 runs under the transfer of chunk c

    (Folding EVERY view on arrival was measured first: 2 917 instead of 3 117 views/s at cfg3 on one GPU -- eight
    141-MB read-modify-write passes per step take more HBM time from the latency-bound binning kernels than the single
    423-MB pass they replace.)  What cannot be hidden is one full-size all-reduce after the last view: every gradient
    element depends on the last backward kernel of the last view, and the optimiser needs the reduced sum before the next
    forward (exposure is reported by ``exposed_ms()``: bench.py prints it next to the stand-alone all-reduce time).

    Works on CPU tensors too (no streams; used by the gloo tests)."""

    def __init__(self, rows, frame_parallel=None, chunks=4, tail=1):
        if rows.dim() != 2:
            raise ValueError("rows must be [views, floats_per_view]")
        self.rows, self.fp = rows, frame_parallel
        self.k, self.n = rows.shape
        self.cuda = rows.is_cuda
        # one view per step: the row IS the sum (no fold, no copy)
        self.acc = rows[0] if self.k == 1 else torch.empty(self.n, dtype=rows.dtype, device=rows.device)
        self.head = max(1, self.k - max(1, int(tail))) if self.k > 1 else 1      # views folded in the first pass
        world = 1 if frame_parallel is None else frame_parallel.world
        c = max(1, int(chunks)) if world > 1 else 1
        step = -(-self.n // c)
        step = -(-step // (world * 64)) * (world * 64)           # chunk boundaries the rs_ag schedule divides, 256-B aligned
        self.bounds = [(lo, min(lo + step, self.n)) for lo in range(0, self.n, step)]
        if self.cuda:
            self.comm = torch.cuda.Stream(rows.device)
            self._ev = [torch.cuda.Event() for _ in range(self.k)]
            self._t = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self._next = 0
        self._works = []
        self._timed = False

    # -- per step -----------------------------------------------------------------------------------------------------
    def begin(self):
        """Call on the stream the step was forked from, before the first view is launched."""
        self._next = 0
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream(self.rows.device))

    def view_done(self, v):
        """Call on the stream view v ran on, right after its backward was queued; views must be reported in order."""
        if v != self._next:
            raise RuntimeError(f"views must be reported in order (expected {self._next}, got {v})")
        self._next += 1
        if not self.cuda:
            self._fold(v)
            return
        self._ev[v].record(torch.cuda.current_stream(self.rows.device))
        if v < self.head - 1:
            return                                               # folded together with the rest of the head
        with torch.cuda.stream(self.comm):
            for u in (range(self.head) if v == self.head - 1 else (v,)):
                self.comm.wait_event(self._ev[u])
            if v == self.k - 1 and self._timed:
                self._t[0].record(self.comm)                     # = the moment the last view's gradients exist
            self._fold(v)

    def _fold(self, v):
        last = v == self.k - 1
        if self.k == 1:
            pass                                                 # acc is rows[0]
        elif v < self.head - 1:
            return
        elif v == self.head - 1:
            if self.head == 1:
                self.acc.copy_(self.rows[0])
            else:
                torch.sum(self.rows[:self.head], dim=0, out=self.acc)
        elif not last:
            self.acc.add_(self.rows[v])
        if not last:
            return
        for lo, hi in self.bounds:
            a = self.acc[lo:hi]
            if self.k > 1 and self.head < self.k:
                a.add_(self.rows[v][lo:hi])
            if self.fp is not None:
                if self.cuda:                                    # queued behind this fold; the next fold does not wait for it
                    self._works += self.fp.all_reduce_grads(a, async_op=True)
                else:
                    self.fp.all_reduce_grads(a)

    def finish(self):
        """Join: the caller's current stream waits for the reduced sum.  Returns ``acc``."""
        if self._next != self.k:
            raise RuntimeError(f"only {self._next} of {self.k} views were reported")
        if self.cuda:
            with torch.cuda.stream(self.comm):
                if self.fp is not None:
                    self.fp.wait(self._works, None)
                    if self.fp.average and self.fp.world > 1:
                        self.acc.div_(self.fp.world)
                self._works = []
                if self._timed:
                    self._t[1].record(self.comm)
            torch.cuda.current_stream(self.rows.device).wait_stream(self.comm)
        return self.acc

    def one_shot(self):
        """Reference schedule: the same folds in the same order after ALL views are done, then ONE collective over the whole
        buffer (what the pipelined schedule must reproduce bit for bit; the pre-round-2 behaviour of bench.py)."""
        if self.k > 1:
            if self.head == 1:
                self.acc.copy_(self.rows[0])
            else:
                torch.sum(self.rows[:self.head], dim=0, out=self.acc)
            for v in range(self.head, self.k):
                self.acc.add_(self.rows[v])
        if self.fp is not None:
            self.fp.all_reduce_grads(self.acc)
        return self.acc

    # -- measurement --------------------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._timed = bool(on) and self.cuda

    def exposed_ms(self):
        """After a synchronised step run with timing enabled: time from "last view's gradients exist" to "reduced sum
        ready" on the communication stream = the part of the reduction nothing can hide."""
        return self._t[0].elapsed_time(self._t[1])
