#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of the one-view-per-step cfg3 bench + timeline.  Usage: bash tools/prof_k1.sh <tag> [extra bench args]
TAG=${1:-k1}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $ROOT/bench.py --steps 40 --warmup 5 --views-per-step 1 --streams 1 --no-cpu-baseline "$@" > $OUT/c3.log 2>&1
f=$(find $OUT/c3 -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/timeline.py $f $OUT/timeline.csv
tail -1 $OUT/c3.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +2M -delete
