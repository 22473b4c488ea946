#!/bin/bash
# train step: where does the rasteriser forward start relative to the decode's end?  variants via env (experiment)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/step_order; mkdir -p $OUT
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python3 $ROOT/bench.py --workload train --steps 200 --warmup 20 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', j['ms_per_step'])"
}
trace() {
  name=$1; shift
  ( export "$@"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/t_$name.log 2>&1 )
  python3 $ROOT/tools/step_trace.py $(find $OUT/t_$name -name "*kernel_trace.csv" | head -1) > $OUT/trace_$name.log 2>&1
  rm -rf $OUT/t_$name
}
run base X=1; run noreg SINGS_STEP_NOREG=1; run rfirst SINGS_STEP_ORDER=raster_first
run base2 X=1; run rfirst2 SINGS_STEP_ORDER=raster_first
trace noreg SINGS_STEP_NOREG=1; trace rfirst SINGS_STEP_ORDER=raster_first
