#!/bin/bash
# kernel trace of one graph-replayed training step: bash tools/r04_prof_train.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04_train; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -c 600 $OUT/bench_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tt -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_trace.log 2>&1
python3 $ROOT/tools/step_trace.py $(find $OUT/tt -name "*kernel_trace.csv" | head -1) > $OUT/train_step_trace.log 2>&1
find $OUT/tt -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/train_step_trace.log | head -120
