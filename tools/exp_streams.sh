#!/bin/bash
# GPU box: views-per-step / streams sweep of the default raster workload:  bash tools/exp_streams.sh "1 1" "4 2" ...
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/exp_streams; mkdir -p $OUT
for cfg in "$@"; do
  set -- $cfg
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --views-per-step $1 --streams $2 > $OUT/k$1_s$2.json 2> $OUT/k$1_s$2.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/k$1_s$2.json").read().strip().splitlines()[-1]); print("k=$1 s=$2", round(d["value"],1), "views/s", round(d["ms_per_step"],4), "ms/step")
except Exception as e:
    print("k=$1 s=$2 FAILED", e); print(open("$OUT/k$1_s$2.err").read()[-2000:])
PY
done
