"""bench.py's command line contract, run as the driver runs it (child processes, last stdout line = JSON)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gaussians", "20000", "--width", "640", "--height", "368", "--steps", "3", "--warmup", "1"]


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e,
                       timeout=timeout, cwd=ROOT)
    return p


def _line(p):
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_single_gpu_line_carries_the_contract():
    j = _line(_bench("--gpus", "1", *SMALL))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "roofline_hbm", "cpu_baseline", "train_step_ms_one_view"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0
    assert j["unit"] == "views/s" and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] in (cb["usable_cores"], 16) and cb["usable_cores"] <= os.cpu_count()
    assert set(cb["median_ms_by_points"]) == {"6890", "50000", "200000"}
    assert cb["raster_oracle_1core"]["cores"] == 1


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_and_the_collective_layer_sees_them():
    """`python bench.py --gpus 2` must really run two ranks (round 1 ignored the flag).  A one-GPU box is oversubscribed:
    both ranks share the device and the collectives are host-staged gloo (RCCL refuses two ranks on one device); on a
    box with >= 2 GPUs the same command runs over RCCL and `rccl_world` is 2."""
    import torch
    j = _line(_bench("--gpus", "2", "--no-cpu-baseline", *SMALL))
    assert j["n_gpus"] == 2 and j["dist_world"] == 2
    assert j["config"]["parallelism"] == "dp2"
    if torch.cuda.device_count() >= 2:
        assert j["rccl_world"] == 2 and j["dist_backend"] == "nccl" and j["ranks_per_device"] == 1
    else:
        assert j["rccl_world"] is None and j["dist_backend"] == "gloo" and j["ranks_per_device"] == 2
    assert j["allreduce_bytes"] == 20000 * 59 * 4 and j["allreduce_ms"] > 0 and j["allreduce_exposed_ms"] > 0
    assert j["value"] > 0


@pytest.mark.gpu
def test_world_size_mismatch_is_an_error():
    p = _bench("--gpus", "2", *SMALL, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stdout + p.stderr)


def test_gpus_flag_spawns_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert p.returncode != 0
    assert "rank exit codes" in p.stderr and "no CPU path" in p.stderr
