#!/bin/bash
# Usage (on the GPU box, through gpurun): bash tools/gpu_check.sh <tag> [pytest args...]
# Runs the GPU test suite, the default bench line, the one-view bench and the 2-rank bench, every step under its own timeout;
# everything lands in gpurun_out/<tag>/.
tag=${1:-check}; shift
out=gpurun_out/$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --timeout=420 "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/rc.txt
tail -25 $out/pytest.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?" | tee -a $out/rc.txt
tail -c 1500 $out/bench_default.err; python - <<PY
import json
try:
    j = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
    print({k: j[k] for k in ("value", "ms_per_step", "ms_per_view", "train_step_ms_one_view", "kernel_ms")})
    print(j["roofline"]); print(j.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
timeout 400 python bench.py --gpus 2 --no-cpu-baseline --steps 50 > $out/bench_g2.json 2> $out/bench_g2.err; echo "g2 rc=$?" | tee -a $out/rc.txt
tail -c 600 $out/bench_g2.err; tail -c 1200 $out/bench_g2.json
