#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3 summaries that profiles/ keeps.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01d'
# rocprofv3 gets `python3 <script>` directly after `--` (no wrappers), counters in their own passes.
# "k1" = --views-per-step 1 --streams 1: one view at a time on one stream, so a kernel's rocprof duration is its own
# (the default bench overlaps the views of a step on two streams: kernels of different views share the GPU and every
# one of them takes longer while the step gets shorter).
set -u
TAG=${1:-r01d}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K1="--views-per-step 1 --streams 1"
python3 $ROOT/bench.py --steps 50 --warmup 10 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 $ROOT/bench.py --steps 50 --warmup 10 $K1 --no-cpu-baseline > $OUT/bench_cfg3_k1.json 2> $OUT/bench_cfg3_k1.err
python3 $ROOT/bench.py --workload avatar --steps 50 --warmup 10 > $OUT/bench_avatar.json 2> $OUT/bench_avatar.err
python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 $ROOT/bench.py --steps 30 --warmup 5 $K1 --no-cpu-baseline > $OUT/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3d -o c3d -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c3d.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/av -o av -- python3 $ROOT/bench.py --workload avatar --steps 30 --warmup 5 --no-cpu-baseline > $OUT/av.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/tr.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/pmc_$ctr.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_SQ -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
find $OUT -name "*.csv" | head -40
# keep the merge under the 64 MiB limit: drop per-dispatch traces, keep stats + counter files
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
