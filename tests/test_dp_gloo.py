"""Frame-parallel host logic on CPU: world_size-2 gloo processes."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sings_amd.dp import FrameParallel, FrameSharder, GradientPipeline


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, algorithm, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fp = FrameParallel(algorithm=algorithm)
        n = 1003                                       # not a multiple of world: exercises padding
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        fp.all_reduce_grads(g)
        acc = torch.full((50,), float(rank + 1)); den = torch.ones(50); rad = torch.arange(50, dtype=torch.int32) * (rank + 1)
        fp.reduce_densification_stats(acc, den, rad)
        p = torch.full((7,), float(rank))
        fp.broadcast_([p])
        loss = fp.reduce_scalar(float(rank), "mean")
        sh = FrameSharder(71, world, rank, seed=5)
        out.put((rank, g.numpy(), acc.numpy(), den.numpy(), rad.numpy(), p.numpy(), loss,
                 [sh.frame(t) for t in range(80)], sh.eval_frames()))
    finally:
        dist.destroy_process_group()


def _run(algorithm):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, algorithm, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _check(res):
    exp = np.arange(1003, dtype=np.float32) * 3
    for r in res:
        assert np.array_equal(r[1], exp)
        assert (r[2] == 3).all() and (r[3] == 2).all()
        assert np.array_equal(r[4], np.arange(50) * 2)
        assert (r[5] == 0).all()
        assert abs(r[6] - 0.5) < 1e-12
    # per step the two ranks render different frames; over one epoch every frame is rendered once
    f0, f1 = res[0][7], res[1][7]
    assert all(a != b for a, b in zip(f0, f1))
    epoch = [x for pair in zip(f0, f1) for x in pair][:71]
    assert sorted(epoch) == list(range(71))
    assert sorted(res[0][8] + res[1][8]) == list(range(71))


def test_all_reduce_world2():
    _check(_run("all_reduce"))


def test_reduce_scatter_all_gather_world2():
    _check(_run("rs_ag"))


def test_sharder_is_deterministic_and_rank_consistent():
    a = FrameSharder(120, 8, 3, seed=1)
    assert [a.frame(t) for t in range(40)] == [FrameSharder(120, 8, 3, seed=1).frame(t) for t in range(40)]
    for t in range(30):
        fr = a.frames_of_step(t)
        assert len(set(fr)) == 8 and fr[3] == a.frame(t)


def test_batched_steps_cover_every_frame_once():
    """bench.py / ViewBatch: rank r renders frames FrameSharder.frame(K * t + v), v < K, at step t (K views per rank and
    step): within an epoch of 120 frames every (rank, step, view) slot is a different frame and all frames are rendered."""
    W, K, F = 8, 3, 120
    sh = [FrameSharder(F, W, r, seed=4) for r in range(W)]
    seen = []
    for t in range(F // (W * K)):
        step = [sh[r].frame(K * t + v) for r in range(W) for v in range(K)]
        assert len(set(step)) == W * K
        seen += step
    assert sorted(seen) == list(range(F))


def _pipeline_worker(rank, world, port, algorithm, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        for k, n, chunks in ((1, 1003, 4), (5, 1003, 4), (8, 4099, 3), (3, 130, 7), (2, 77, 2), (6, 513, 1)):
            g = torch.Generator().manual_seed(100 * rank + k)
            rows = torch.randn((k, n), generator=g) * torch.logspace(-3, 3, n)        # fp32 sums that depend on the order
            fp = FrameParallel(algorithm=algorithm)
            ref = GradientPipeline(rows.clone(), fp, chunks=chunks).one_shot().clone()
            live = rows.clone()
            pipe = GradientPipeline(live, fp, chunks=chunks)
            for _ in range(2):                                                        # two steps through the same object
                live.copy_(rows)                                                      # (every backward rewrites its row)
                got = pipe.reduce().clone()
            res.append((k, n, ref.numpy(), got.numpy(), rows.numpy(), len(pipe.bounds)))
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def _run_pipeline(algorithm):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, algorithm, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for case in range(len(res[0])):
        k, n, ref0, got0, rows0, nchunks = res[0][case]
        _, _, ref1, got1, rows1, _ = res[1][case]
        # chunk-wise fold + chunk-wise collectives == fold everything, then one collective: bit for bit
        assert np.array_equal(ref0, got0) and np.array_equal(ref1, got1)
        assert np.array_equal(got0, got1)                                            # both ranks hold the same sum
        fold = lambda rows: torch.sum(torch.from_numpy(rows), dim=0).numpy() if k > 1 else rows[0]
        assert np.array_equal(got0, fold(rows0) + fold(rows1))                       # inside a rank, then the ranks
        assert nchunks >= 1


def test_pipelined_reduction_matches_one_shot_all_reduce():
    _run_pipeline("all_reduce")


def test_pipelined_reduction_matches_one_shot_rs_ag():
    _run_pipeline("rs_ag")


def _force_worker(port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = {"all_reduce": 0, "reduce_scatter_tensor": 0, "all_gather_into_tensor": 0}
        for name in calls:
            def wrap(fn, name=name):
                def inner(*a, **k):
                    calls[name] += 1
                    return fn(*a, **k)
                return inner
            setattr(dist, name, wrap(getattr(dist, name)))
        res = {}
        rows = torch.randn(3, 1003)
        for algo in ("all_reduce", "rs_ag"):
            lazy = FrameParallel(algorithm=algo)                       # world 1, not forced: returns before the backend
            assert not lazy.active
            before = dict(calls)
            g = rows[0].clone(); lazy.all_reduce_grads(g)
            assert calls == before and torch.equal(g, rows[0])
            fp = FrameParallel(algorithm=algo, force=True)
            assert fp.active
            g = rows[0].clone(); fp.all_reduce_grads(g)
            pipe = GradientPipeline(rows.clone(), fp, chunks=4)
            assert len(pipe.bounds) == 4                               # chunked although there is one rank
            acc = pipe.reduce().clone()
            res[algo] = (torch.equal(g, rows[0]), torch.equal(acc, rows.sum(0)), torch.equal(pipe.one_shot(), rows.sum(0)))
        out.put((res, calls))
    finally:
        dist.destroy_process_group()


def test_forced_one_rank_group_issues_every_collective():
    """FrameParallel(force=True): the collectives of both schedules run in a ONE-rank group (what bench.py does on a single-GPU
    box under SINGS_BENCH_FORCE_DIST=1 so that the nccl branches execute) and are the identity."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(_free_port(), q))
    p.start()
    res, calls = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert res == {"all_reduce": (True, True, True), "rs_ag": (True, True, True)}
    assert calls["all_reduce"] >= 6 and calls["reduce_scatter_tensor"] >= 5 and calls["all_gather_into_tensor"] >= 5


def _packed_worker(rank, world, port, algorithm, out):
    """The frame-parallel collective over the PREFIX of the gradient rows that carries gradient (engines with sh_planar: the SH
    planes not in use stay zero behind it) against the same sum over the full rows in the reference's [P,16,3] layout."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M, deg, k = 517, 16, 0, 3
        nc = (deg + 1) ** 2
        g = torch.Generator().manual_seed(7 + rank)
        head = torch.randn((k, P * 7), generator=g) * torch.logspace(-3, 3, P * 7)       # xyz 3, scales 3, opacity 1
        sh_used = torch.randn((k, nc, P, 3), generator=g)                                 # the planes in use
        # reference layout: [head | sh as [P, M, 3]] -- 45 of the 55 floats per Gaussian are zero
        sh_rows = torch.zeros((k, P, M, 3)); sh_rows[:, :, :nc, :] = sh_used.permute(0, 2, 1, 3)
        full = torch.cat([head, sh_rows.reshape(k, -1)], 1).contiguous()
        # planar layout: [head | sh as [M, P, 3]]: the gradient is the prefix
        sh_planes = torch.zeros((k, M, P, 3)); sh_planes[:, :nc] = sh_used
        packed = torch.cat([head, sh_planes.reshape(k, -1)], 1).contiguous()
        active = P * 7 + nc * P * 3
        fp = FrameParallel(algorithm=algorithm)
        ref = GradientPipeline(full, fp, chunks=4).reduce().clone()
        pipe = GradientPipeline(packed, fp, chunks=4, active=active)
        got = pipe.reduce().clone()
        out.put((rank, ref.numpy(), got.numpy(), active, P, M, nc))
    finally:
        dist.destroy_process_group()


def test_packed_gradient_prefix_equals_the_full_row_sum_world2():
    """only the floats that carry gradient are folded and all-reduced (avatar: 10 of 55 per Gaussian).  World
    size 2 over gloo, both schedules: the reduced prefix equals, bit for bit, the corresponding elements of the full-row sum in
    the reference's layout, and everything behind the prefix is zero."""
    for algorithm in ("all_reduce", "rs_ag"):
        world = 2
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_packed_worker, args=(r, world, port, algorithm, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        for rank, ref, got, active, P, M, nc in res:
            assert active == P * 10 and got.shape == ref.shape
            assert np.array_equal(got[:P * 7], ref[:P * 7])
            ref_sh = ref[P * 7:].reshape(P, M, 3)
            got_sh = got[P * 7:].reshape(M, P, 3)
            assert np.array_equal(got_sh[:nc].transpose(1, 0, 2), ref_sh[:, :nc])
            assert not got[active:].any() and not ref_sh[:, nc:].any()
        assert np.array_equal(res[0][2], res[1][2])


def test_pipeline_active_prefix_follows_the_sh_degree():
    """GradientPipeline.set_active: the folded / reduced prefix of the gradient rows can grow when the SH degree is raised
    (oneupSHdegree, gs_trainer.py:436-438) -- planes behind the old prefix are no longer dropped -- and shrink again (the tail of
    the accumulator is zeroed, not left with the wider fold's sums)."""
    rows = torch.arange(3 * 40, dtype=torch.float32).view(3, 40)
    pipe = GradientPipeline(rows, None, active=10)
    acc = pipe.reduce()
    assert torch.equal(acc[:10], rows[:, :10].sum(0)) and float(acc[10:].abs().max()) == 0
    pipe.set_active(25)
    acc = pipe.reduce()
    assert torch.equal(acc[:25], rows[:, :25].sum(0)) and float(acc[25:].abs().max()) == 0
    assert pipe.bounds[-1][1] == 25
    pipe.set_active(10)
    acc = pipe.reduce()
    assert torch.equal(acc[:10], rows[:, :10].sum(0)) and float(acc[10:].abs().max()) == 0
    import pytest
    with pytest.raises(ValueError):
        pipe.set_active(41)
