#!/bin/bash
# On the GPU box: per-kernel timeline of ONE graph-replayed training step (bench.py --workload train).  Usage: bash tools/trace_train.sh <tag> [bench args]
TAG=${1:-tt}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/train_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/train_trace -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/train_trace.log 2>&1
python3 $ROOT/tools/step_trace.py $(find $OUT/train_trace -name "*kernel_trace.csv" | head -1) > $OUT/train_step_trace.log 2>&1
rm -rf $OUT/train_trace
head -130 $OUT/train_step_trace.log
