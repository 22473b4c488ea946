#!/bin/bash
# GPU box: per-kernel times of the tri-plane backward for one cloud (avatar | uniform)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for c in ${CLOUDS:-avatar uniform}; do
  rm -rf $R/gpurun_out/prof_tp_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tp_$c -o tp -- python3 $R/tools/tp_bwd_time.py $c 2>/dev/null | grep backward
  python3 $R/tools/kstats.py $R/gpurun_out/prof_tp_$c/tp_kernel_stats.csv ${ROWS:-12} sg_
  find $R/gpurun_out/prof_tp_$c -name "*kernel_trace.csv" -delete
done
