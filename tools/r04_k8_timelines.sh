#!/bin/bash
# the two K-frame timelines of tools/collect_profiles.sh alone (8 frames / cameras per dispatch, one stream)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K8="--views-per-step 8 --frames-per-launch 8 --streams 1"
for w in cfg3 avatar; do
  extra=""; [ $w = avatar ] && extra="--workload avatar"
  rm -rf $OUT/${w}_k8t
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_k8t -o $w -- python3 $ROOT/bench.py --steps 20 --warmup 5 $K8 --no-cpu-baseline $extra > $OUT/${w}_k8.log 2>&1
  f=$(find $OUT/${w}_k8t -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/timeline.py $f $OUT/${w}_k8_timeline.csv must=sg_record_sums_kernel
  find $OUT/${w}_k8t -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_k8_kernel_stats.csv \;
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
