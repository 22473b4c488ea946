#!/bin/bash
# the training-step files of the profile campaign only (after a change that touches nothing but the decode / step code):
#   gpurun -- 'bash tools/r05_train_profiles.sh r05'  then  bash tools/r05_train_profiles.sh r05 keep
TAG=${1:-r05}
if [ "${2:-}" = keep ]; then
  P=gpurun_out/prof_$TAG
  for f in bench_train bench_train_k16; do cp $P/$f.json profiles/${TAG}_$f.json; done
  cp $P/train_kernel_stats.csv profiles/${TAG}_train_kernel_stats.csv; cp $P/train_step_trace.log profiles/${TAG}_train_step_trace.log
  exit 0
fi
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TO="timeout 300"
$TO python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
$TO python3 $ROOT/bench.py --workload train --views-per-step 16 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_train_k16.json 2> $OUT/bench_train_k16.err
rm -rf $OUT/tr $OUT/train_trace
$TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 --eager > $OUT/tr.log 2>&1
find $OUT/tr -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
$TO rocprofv3 --kernel-trace --output-format csv -d $OUT/train_trace -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 > $OUT/train_trace.log 2>&1
python3 $ROOT/tools/step_trace.py $(find $OUT/train_trace -name "*kernel_trace.csv" | head -1) > $OUT/train_step_trace.log 2>&1
rm -rf $OUT/tr $OUT/train_trace
tail -c 600 $OUT/bench_train.json; echo; tail -c 300 $OUT/bench_train_k16.json
