"""Process group, timed regions and the frame-parallel self-check of bench.py: one process per GPU (RCCL = the "nccl" backend), barrier +
synchronize brackets, MAX over ranks; what makes an N > 1 run check itself (ranks_agree, dp_parity, both collectives timed).  No oracle."""
import os
import subprocess
import sys
import time

from .roofline import XGMI_LINK_GBS

_T0 = time.perf_counter()
# SINGS_BENCH_FORCE_DIST=1: initialise the process group and ISSUE every collective even with one rank (RCCL accepts a one-rank
# communicator), so that the nccl branches -- init_process_group("nccl", device_id), async all-reduce / in-place reduce-scatter +
# all-gather on device views, work handles, the device-side MAX of the timed region -- run on a single-GPU box
# (tests/test_gpu_bench.py); the line then carries rccl_world = 1 and the allreduce_* keys.
FORCE_DIST = bool(os.environ.get("SINGS_BENCH_FORCE_DIST"))


# one schema for every N: the collective keys are present (null) when no collective ran
COMM_KEYS = ("allreduce_ms", "allreduce_bytes", "allreduce_algorithm", "allreduce_per_link_bound_ms", "allreduce_exposed_ms")
MIN_TIMED_S = 0.5              # the timed region is repeated (whole regions of exactly --steps steps) until it adds up to this
LIGHT_TIMED_S = 0.3            # ... of a secondary leg (--light)
MAX_REPEATS = 5000


def _log(msg):
    """Progress on stderr (stdout carries only the JSON line): where a run is, should it ever stall."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(a, script):
    """`python bench.py --gpus N` (N > 1, not under a launcher): start N rank processes and relay rank 0's line.

    Runs BEFORE anything in this process has initialised the GPU (no HIP call, no torch.cuda.is_available(); torch is not
    even imported yet), and never exec()s: the ranks are ordinary children.  Ranks share a device only when the box has
    fewer GPUs than ranks (single-GPU test boxes); RCCL refuses two ranks on one device, so that oversubscribed mode uses
    host-staged gloo collectives and says so in the JSON line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    import tempfile
    out0 = tempfile.TemporaryFile()
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; if one dies, the others would sit in a collective until its timeout: stop exactly those PIDs
    rcs = [None] * a.gpus
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill(); rcs[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text = out0.read().decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.strip()]
    if any(rcs):
        print(text, file=sys.stderr)
        raise SystemExit(f"bench.py: rank exit codes {rcs}")
    if not lines or not lines[-1].lstrip().startswith("{"):
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    for ln in lines[:-1]:
        print(ln)
    print(lines[-1], flush=True)


def dist_setup(a):
    """One process per GPU.  Returns (rank, world, device, dist module or None, info dict for the JSON line)."""
    import torch
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    ndev = torch.cuda.device_count()
    if ndev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    oversub = world > ndev
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    dist, info = None, {"rccl_world": None, "dist_backend": None, "dist_world": 1, "ranks_per_device": 1}
    if world > 1 or FORCE_DIST:                                    # the env knob exercises the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if oversub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
        # one rank per GPU, or the run is not the measurement it claims to be: every rank reports its device, all must differ
        # (single-GPU test boxes are knowingly oversubscribed: gloo, `ranks_per_device` > 1, said in the line)
        mine = torch.tensor([local_rank % ndev], dtype=torch.int64, device="cpu" if oversub else dev)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        devices = [int(g.item()) for g in got]
        if not oversub and not FORCE_DIST and len(set(devices)) != world:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the ranks sit on devices {devices}: one rank per GPU")
        info = {"rccl_world": None if oversub else dist.get_world_size(), "dist_backend": dist.get_backend(),
                "dist_world": dist.get_world_size(), "ranks_per_device": -(-world // ndev), "rank_devices": devices}
    return rank, world, dev, dist, info


def make_frame_parallel(ctx, nfloats):
    """The collective layer of a leg + which algorithm it runs.  SINGS_DP_ALGO=all_reduce|rs_ag forces one; otherwise, with several
    real ranks, BOTH are timed stand-alone on a scratch buffer of the step's gradient size (10 calls each after 3 warm-ups, device
    events, MAX over ranks so that every rank reaches the same verdict) and the faster one runs the step -- the first real
    multi-GPU run therefore measures the better schedule AND reports both (`allreduce_ms_by_algorithm`).  -> (fp or None, info)."""
    import torch
    rank, world, dev, dist, dinfo = ctx
    if dist is None:
        return None, {}
    from sings_amd.dp import FrameParallel
    staged = dist.get_backend() != "nccl"
    mk = lambda algo: FrameParallel(algorithm=algo, host_staged=staged, force=FORCE_DIST)
    forced = os.environ.get("SINGS_DP_ALGO")
    if forced:
        return mk(forced), {"allreduce_algorithm_chosen_by": "SINGS_DP_ALGO"}
    if world == 1 and not FORCE_DIST:
        return mk("all_reduce"), {"allreduce_algorithm_chosen_by": "default (one rank)"}
    # (a FORCED one-rank nccl group takes the measuring path too: it is the only way to execute it on a one-GPU box)
    scratch = torch.zeros(int(nfloats), dtype=torch.float32, device=dev)
    times = {}
    for algo in ("all_reduce", "rs_ag"):
        f = mk(algo)
        for _ in range(3):
            f.all_reduce_grads(scratch)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f.all_reduce_grads(scratch)
        e1.record()
        torch.cuda.synchronize()
        tt = torch.tensor([e0.elapsed_time(e1) / 10], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times[algo] = float(tt.item())
    del scratch
    best = min(times, key=times.get)
    return mk(best), {"allreduce_ms_by_algorithm": times, "allreduce_algorithm_chosen_by": "measured in this run (the faster of the two, stand-alone)"}


def timed_region(dist, dev, steps, step):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks (seconds)."""
    import torch
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el


def timed_repeats(dist, dev, steps, step, min_s=None):
    """The timed region, repeated: whole regions of EXACTLY `steps` steps (each bracketed by barrier + synchronize, MAX over
    ranks) until they add up to MIN_TIMED_S seconds and there are at least two.  The driver's `--steps 20` is 42 ms of work
    for a 2-ms step -- one region of that length moves by +-1.5 % with the box; the line reports the MEDIAN region (and min /
    max).  Every rank sees the same (reduced) durations, so all ranks take the same number of repetitions."""
    min_s = MIN_TIMED_S if min_s is None else min_s
    els = []
    while True:
        els.append(timed_region(dist, dev, steps, step))
        if (len(els) >= 2 and sum(els) >= min_s) or len(els) >= MAX_REPEATS:
            return els


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def _grad_sha256(t):
    """sha256 of a device buffer's bytes (tests: the forced one-rank RCCL run must reproduce the no-dist gradients bit for bit)."""
    import hashlib
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()


def allreduce_probe(fp, buf, iters=10):
    """Stand-alone collective on the step's gradient buffer: ms per call (device events; MAX over ranks is implied by the
    collective itself), bytes, and the xGMI per-link lower bound 2 (S/W) / 153 GB/s of a reduce-scatter + all-gather that
    uses every link of the fully connected mesh."""
    import torch
    if fp is None or not fp.active:
        return None
    scratch = buf.clone()
    for _ in range(3):
        fp.all_reduce_grads(scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fp.all_reduce_grads(scratch)
    e1.record()
    torch.cuda.synchronize()
    S = buf.numel() * buf.element_size()
    return {"allreduce_ms": e0.elapsed_time(e1) / iters, "allreduce_bytes": S, "allreduce_algorithm": fp.algorithm,
            "allreduce_per_link_bound_ms": 2.0 * (S / fp.world) / (XGMI_LINK_GBS * 1e9) * 1e3}


def usable_cores():
    """Host cores this process may actually run on: the affinity mask and the cgroup CPU quota, not just os.cpu_count()
    (a container can see 256 CPUs and be allowed a fraction of them; an OpenMP team of 256 threads on a 16-CPU quota spends
    its time in barriers)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def densification_stats_check(ctx, fp, engs):
    """Frame-parallel densification statistics (sings_hybrid.py:1013-1015, gs_trainer.py:486-492): every rank accumulates
    |viewspace gradient| (sum), the visibility count (sum) and the largest screen radius (max) of ITS frames; the three are reduced
    so that all ranks take identical densify / prune decisions.  Reduced here from the last step's engines, then compared across
    ranks by hash.  -> keys for the line (the same on every rank)."""
    import hashlib
    import torch
    rank, world, dev, dist, _ = ctx
    e0 = engs[0]
    P = e0.P
    acc = torch.zeros(P, device=dev); den = torch.zeros(P, device=dev); rad = torch.zeros(P, dtype=torch.int32, device=dev)
    for e in engs:
        m2 = e.d_means2D.view(-1, P, 3); r = e.radii.view(-1, P)
        vis = r > 0
        acc += (m2[..., :2].norm(dim=-1) * vis).sum(0); den += vis.sum(0).float(); rad = torch.maximum(rad, r.max(0).values)
    fp.reduce_densification_stats(acc, den, rad)
    h = hashlib.sha256(acc.cpu().numpy().tobytes() + den.cpu().numpy().tobytes() + rad.cpu().numpy().tobytes()).digest()
    return {"densification_stats": {"reduced": True, "ranks_agree": _ranks_agree(ctx, h), "visible_sum": float(den.sum()),
                                    "max_radius": int(rad.max())}}


def _ranks_agree(ctx, digest):
    """all_gather of a 32-byte digest: True iff every rank holds the same bytes."""
    import torch
    rank, world, dev, dist, _ = ctx
    on_dev = dist.get_backend() == "nccl"
    mine = torch.tensor(list(digest), dtype=torch.uint8, device=dev if on_dev else "cpu")
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return all(bool(torch.equal(g, got[0])) for g in got)


def dp_self_check(ctx, step0, render_flat, n_views_world):
    """N > 1 (or a forced one-rank group), outside every timed region -- what makes the first real multi-GPU run decisive:
      ranks_agree   all ranks all_gather the sha256 of their reduced gradient buffer: the collective must leave the SAME bytes
                    everywhere;
      dp_parity     every view of the step is re-rendered ONCE, by the rank that owns it, through the one-view engine
                    (`render_flat(v)` -> that view's gradient in the buffer's layout; rank r owns views r k .. r k + k - 1); the
                    per-rank fp64 sums are added with one fp64 all-reduce and rank 0 compares the total with the reduced sum of the
                    step (rtol 2e-4 + 2e-6 max|g|).  (Rounds 4-5: rank 0 re-rendered all world x k views itself while the others
                    waited -- 128 renders at 8 ranks; now k per rank, in parallel.)
    `step0()` runs one step on every rank and returns the reduced buffer.  Oversubscribed single-GPU test boxes (gloo, host-staged)
    run the same code."""
    import hashlib
    import torch
    rank, world, dev, dist, dinfo = ctx
    acc = step0()
    torch.cuda.synchronize()
    agree = _ranks_agree(ctx, hashlib.sha256(acc.detach().cpu().numpy().tobytes()).digest())
    res = {"ranks_agree": bool(agree), "dp_parity": None}
    got = acc.detach().double().clone()
    k = n_views_world // world
    ref = torch.zeros_like(got)
    for v in range(rank * k, rank * k + k):
        ref += render_flat(v).detach().double()
    torch.cuda.synchronize()
    if dist.get_backend() == "nccl":
        dist.all_reduce(ref)
    else:                                                            # (host-staged gloo on an oversubscribed test box)
        h = ref.cpu()
        dist.all_reduce(h)
        ref = h.to(dev)
    if rank == 0:
        scale = float(ref.abs().max()) + 1e-30
        err = (got - ref).abs()
        bad = int((err > 2e-4 * ref.abs() + 2e-6 * scale).sum())
        res["dp_parity"] = {"ok": bad == 0, "views": n_views_world, "max_rel": float(err.max()) / scale, "violations": bad,
                            "tol": "rtol 2e-4 + 2e-6 x max|g|",
                            "note": "reduced sum of one step vs every view re-rendered once by its own rank through the one-view engine "
                                    "(fp64 sums, one fp64 all-reduce)"}
        del err
    del got, ref
    torch.cuda.synchronize()
    dist.barrier()
    return res


def one_view_step_by_algorithm(ctx, fp, grad_flat, one_view, n, timed_repeats, median):
    """The reference's operating point, one view per optimisation step, with BOTH collectives in the same run: view + all-reduce of its
    gradients, timed like the batched region (MAX over ranks inside timed_repeats).  The model (scaling_model) says 2.8x at 8 GPUs
    with the ring all_reduce and 6.3x with reduce-scatter + all-gather at one view per step: THIS is the number that tests it."""
    import torch
    rank, world, dev, dist, dinfo = ctx
    from sings_amd.dp import FrameParallel
    staged = dist.get_backend() != "nccl"
    out = {}
    for algo in ("all_reduce", "rs_ag"):
        f = FrameParallel(algorithm=algo, host_staged=staged, force=FORCE_DIST)

        def step(_i=0, f=f):
            one_view()
            f.all_reduce_grads(grad_flat)
        for _ in range(5):
            step()
        el = median(timed_repeats(dist, dev, n, step, min_s=0.2))
        out[algo] = {"ms_per_step": el / n * 1e3, "views_per_s": world * n / el}
    return out


def exposed_by_algorithm(ctx, pipe, step):
    """The part of the collective the batched step cannot hide, for BOTH algorithms in the same run (median of 10 synchronised
    steps each; MAX over ranks): the chunked fold + collective of sings_amd.dp.GradientPipeline with its FrameParallel swapped."""
    import torch
    rank, world, dev, dist, dinfo = ctx
    from sings_amd.dp import FrameParallel
    staged = dist.get_backend() != "nccl"
    keep = pipe.fp
    out = {}
    pipe.enable_timing(True)
    for algo in ("all_reduce", "rs_ag"):
        pipe.fp = FrameParallel(algorithm=algo, host_staged=staged, force=FORCE_DIST)
        ex = []
        for _ in range(10):
            torch.cuda.synchronize(); dist.barrier()
            step()
            torch.cuda.synchronize()
            ex.append(pipe.exposed_ms())
        tt = torch.tensor([sorted(ex)[len(ex) // 2]], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out[algo] = float(tt.item())
    pipe.enable_timing(False)
    pipe.fp = keep
    return out
