#!/bin/bash
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/graph_env; mkdir -p $OUT
run() {  # name, flags, env...
  name=$1; flags=$2; shift; shift
  env "$@" timeout 300 python3 $ROOT/bench.py --workload train --steps 200 --warmup 20 --no-cpu-baseline $flags > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', j['ms_per_step'])" || tail -5 $OUT/$name.err
}
trace() {
  name=$1; shift
  ( export "$@"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/t_$name.log 2>&1 )
  python3 $ROOT/tools/step_trace.py $(find $OUT/t_$name -name "*kernel_trace.csv" | head -1) > $OUT/trace_$name.log 2>&1
  rm -rf $OUT/t_$name
}
run base "" X=1
for h in 2 8 16; do run hwq$h "" GPU_MAX_HW_QUEUES=$h; done
run hwq8_gq6 "" GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=6
run hwq8_gq8 "" GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
trace hwq8 GPU_MAX_HW_QUEUES=8
