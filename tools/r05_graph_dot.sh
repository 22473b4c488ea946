#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/graph_dot; mkdir -p $OUT; cd $OUT
SINGS_TORCH_PROFILE=1 timeout 300 python3 $ROOT/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --eager > prof.json 2> prof.err
grep "^ATEN" prof.err > aten.txt; wc -l aten.txt
