#!/bin/bash
# The captured training step under the HIP runtime's graph-executor knobs, its node -> queue assignment and one traced step
# (profiles/r05_graph_queues.log):   gpurun -- 'bash tools/r05_graph_env.sh'
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/graph_env; mkdir -p $OUT
run() {  # name, bench flags, env assignments...
  name=$1; flags=$2; shift; shift
  env "$@" timeout 300 python3 $ROOT/bench.py --workload train --steps 200 --warmup 20 --no-cpu-baseline $flags > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "import json,sys; j=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', j['ms_per_step'])" || tail -5 $OUT/$name.err
}
trace() {  # name, env assignments... (rocprofv3 gets python3 directly: no env / bash hop after the GPU is initialised)
  name=$1; shift
  ( export "$@"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/t_$name.log 2>&1 )
  python3 $ROOT/tools/step_trace.py $(find $OUT/t_$name -name "*kernel_trace.csv" | head -1) > $OUT/trace_$name.log 2>&1
  rm -rf $OUT/t_$name
}
run base "" X=1
for q in 2 3 5; do run queues$q "" DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; done
for h in 2 8; do run hwq$h "" GPU_MAX_HW_QUEUES=$h; done
run k16 "--views-per-step 16" X=1
run eager "--eager" X=1
trace base X=1
# the instantiated graph with a StreamId per node (written into the working directory)
mkdir -p $OUT/dot; cd $OUT/dot
DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 300 python3 $ROOT/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline > run.json 2> run.err
