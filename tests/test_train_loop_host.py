"""CPU: the decision logic of sings_amd.train_loop (densify / prune from the accumulated statistics: sings_hybrid.py:968-1004) and its
frame-parallel contract -- the statistics are reduced (sum, sum, max) BEFORE the decision, so every rank takes the same one."""
import hashlib
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def test_densify_decision_follows_the_reference_rules():
    from sings_amd.train_loop import densify_decision
    accum = torch.tensor([[0.0], [4.0], [9.0], [1.0], [0.0], [6.0]])
    denom = torch.tensor([[0.0], [2.0], [3.0], [4.0], [5.0], [2.0]])               # row 0: 0 / 0 = NaN -> 0 (sings_hybrid.py:987-988)
    scales = torch.tensor([[0.1], [0.1], [0.5], [0.1], [0.1], [0.05]]).repeat(1, 3)
    opacity = torch.tensor([[0.9], [0.9], [0.9], [0.01], [0.9], [0.002]])
    clone, prune = densify_decision(accum, denom, scales, opacity, max_grad=2.0, scale_threshold=0.2, min_opacity=0.005, big_scale=0.4)
    # grads = [0, 2, 3, .25, 0, 3]; clone: grad >= 2 and max scale <= 0.2 -> rows 1 and 5 (row 2 is too large)
    assert clone.tolist() == [False, True, False, False, False, True]
    # prune on the extended set (6 + 2 clones of rows 1, 5): opacity < 0.005 -> row 5 and its clone; scale > 0.4 -> row 2
    assert prune.tolist() == [False, False, True, False, False, True, False, True]
    # a pure function: same inputs, same outputs; float64 statistics decide the same
    c2, p2 = densify_decision(accum.double(), denom.double(), scales, opacity, 2.0, 0.2, 0.005, 0.4)
    assert torch.equal(c2, clone) and torch.equal(p2, prune)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from sings_amd.dp import FrameParallel
    from sings_amd.train_loop import densify_decision
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5000
        g = torch.Generator().manual_seed(100 + rank)                              # every rank saw OTHER frames: other statistics
        accum = torch.rand((n, 1), generator=g) * 3e-4 * 10
        denom = torch.randint(0, 11, (n, 1), generator=g).float()
        radii = torch.randint(0, 40, (n,), generator=g).float()
        gs = torch.Generator().manual_seed(7)                                      # the replicated model: same attributes everywhere
        scales = torch.rand((n, 3), generator=gs) * 0.02
        opacity = torch.rand((n, 1), generator=gs)
        local = densify_decision(accum, denom, scales, opacity, 2e-4, 0.01, 0.05, 1.0)
        fp = FrameParallel()
        fp.reduce_densification_stats(accum, denom, radii)
        red = densify_decision(accum, denom, scales, opacity, 2e-4, 0.01, 0.05, 1.0)
        h = lambda d: hashlib.sha256(d[0].numpy().tobytes() + d[1].numpy().tobytes()).hexdigest()
        out.put((rank, h(local), h(red), int(red[0].sum()), int(red[1].sum()), float(denom.max())))
    finally:
        dist.destroy_process_group()


def test_world2_ranks_take_the_same_decision_from_the_reduced_statistics():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, r0, nc0, np0, dmax0), (_, l1, r1, nc1, np1, dmax1) = res
    assert l0 != l1, "the local statistics differ, and so would the local decisions"
    assert r0 == r1 and (nc0, np0) == (nc1, np1) and nc0 > 0 and np0 > 0
    assert dmax0 == dmax1 > 10                                                     # (the visibility counts were summed)


def test_arena_sync_refuses_a_stale_registration():
    import pytest
    from sings_amd import decode
    try:
        a, b = torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5))
        flat = torch.zeros(17)
        views = decode.set_gradient_arena([a, b], flat)
        a.grad, b.grad = torch.ones(4, 3), torch.ones(5)
        assert decode.arena_sync([a, b], views, True) == 2 and float(flat.sum()) == 17.0
        decode.invalidate_gradient_arena()                                         # what AvatarStep.set_topology does
        with pytest.raises(RuntimeError, match="stale"):
            decode.arena_sync([a, b], views, True)
        a2 = torch.nn.Parameter(torch.zeros(6, 3))
        views2 = decode.set_gradient_arena([a2, b], torch.zeros(23))
        with pytest.raises(RuntimeError, match="stale"):
            decode.arena_sync([a, b], views2, True)                                # the OLD parameter list against the new registration
        a2.grad = torch.ones(6, 3)
        assert decode.arena_sync([a2, b], views2, True) == 2
    finally:
        decode.set_gradient_arena(None, None)
