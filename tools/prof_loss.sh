#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the photometric-loss kernels alone (tools/loss_time.py).  Usage: bash tools/prof_loss.sh <tag>
TAG=${1:-loss}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l -o l -- python3 $ROOT/tools/loss_time.py "$@" > $OUT/l.log 2>&1
f=$(find $OUT/l -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 $ROOT/tools/kstats.py $f 12
cat $OUT/l.log | tail -4
find $OUT -name "*kernel_trace.csv" -size +2M -delete
