#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3 summaries that profiles/ keeps.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01c'
# rocprofv3 gets `python3 <script>` directly after `--` (no wrappers), counters in their own passes.
set -u
TAG=${1:-r01c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 50 --warmup 10 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 $ROOT/bench.py --workload avatar --steps 50 --warmup 10 > $OUT/bench_avatar.json 2> $OUT/bench_avatar.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/av -o av -- python3 $ROOT/bench.py --workload avatar --steps 30 --warmup 5 --no-cpu-baseline > $OUT/av.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_$ctr.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_SQ -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
find $OUT -name "*.csv" | head -40
# keep the merge under the 64 MiB limit: drop per-dispatch traces, keep stats + counter files
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
