#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3 summaries that profiles/ keeps.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r04'
# then, in the build container:  python tools/pmc_summary.py gpurun_out/prof_r04 r04
#                                python tools/pmc_summary.py gpurun_out/prof_r04/avatar r04_avatar workload=avatar gaussians=150000 width=512 height=896 sh_degree=0
#                                python tools/pmc_summary.py gpurun_out/prof_r04/k8 r04_k8 frames_per_launch=8
#                                python tools/pmc_summary.py gpurun_out/prof_r04/avatar_k8 r04_avatar_k8 workload=avatar gaussians=150000 width=512 height=896 sh_degree=0 frames_per_launch=8
# every program runs under its own `timeout` (a kernel fault leaves the process hanging in the core-dump handler on this pool)
# rocprofv3 gets `python3 <script>` directly after `--` (no wrappers), counters in their own passes (never together with a trace).
# "k1" = --views-per-step 1 --streams 1: one view at a time on one stream, so a kernel's rocprof duration is its own
# (the default bench overlaps the views of a step on three streams: kernels of different views share the GPU and every
# one of them takes longer while the step gets shorter).
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $OUT/avatar $OUT/k8 $OUT/avatar_k8
cd /tmp && export TMPDIR=/tmp
K1="--views-per-step 1 --streams 1"
K8="--views-per-step 8 --frames-per-launch 8 --streams 1"       # 8 frames / cameras per dispatch, one stream: a kernel's duration is its own
TO="timeout 300"
$TO python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
$TO python3 $ROOT/bench.py --steps 100 --warmup 10 $K1 --no-cpu-baseline > $OUT/bench_cfg3_k1.json 2> $OUT/bench_cfg3_k1.err
$TO python3 $ROOT/bench.py --workload avatar --steps 100 --warmup 10 > $OUT/bench_avatar.json 2> $OUT/bench_avatar.err
$TO python3 $ROOT/bench.py --workload avatar --steps 100 --warmup 10 $K1 --no-cpu-baseline > $OUT/bench_avatar_k1.json 2> $OUT/bench_avatar_k1.err
$TO python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/bench_train.json 2> $OUT/bench_train.err
$TO python3 $ROOT/bench.py --workload train --views-per-step 16 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_train_k16.json 2> $OUT/bench_train_k16.err
$TO python3 $ROOT/bench.py --gaussians 50000 --width 512 --height 512 --sh-degree 0 --forward-only --steps 200 --no-cpu-baseline > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
$TO python3 $ROOT/bench.py --gaussians 500000 --width 2048 --height 2048 --regularisers --steps 40 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
SINGS_BENCH_FORCE_DIST=1 $TO python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
SINGS_BENCH_FORCE_DIST=1 SINGS_DP_ALGO=rs_ag $TO python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_rccl_world1_rs_ag.json 2> $OUT/bench_rccl_world1_rs_ag.err
$TO python3 $ROOT/tools/wrapper_time.py > $OUT/wrapper_time.log 2>&1
# kernel traces (durations + gaps)
for w in cfg3 avatar; do
  extra=""; [ $w = avatar ] && extra="--workload avatar"
  $TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o $w -- python3 $ROOT/bench.py --steps 40 --warmup 5 $K1 --no-cpu-baseline $extra > $OUT/$w.log 2>&1
  f=$(find $OUT/$w -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/timeline.py $f $OUT/${w}_k1_timeline.csv > /dev/null 2>&1
  find $OUT/$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_k1_kernel_stats.csv \;
done
# K frames / cameras per launch on ONE stream: the K-frame kernels' own durations (the default bench overlaps two such batches)
for w in cfg3 avatar; do
  extra=""; [ $w = avatar ] && extra="--workload avatar"
  $TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_k8t -o $w -- python3 $ROOT/bench.py --steps 20 --warmup 5 $K8 --no-cpu-baseline $extra > $OUT/${w}_k8.log 2>&1
  f=$(find $OUT/${w}_k8t -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/tools/timeline.py $f $OUT/${w}_k8_timeline.csv must=sg_record_sums_kernel > /dev/null 2>&1
  find $OUT/${w}_k8t -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_k8_kernel_stats.csv \;
done
$TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/avd -o avd -- python3 $ROOT/bench.py --workload avatar --steps 30 --warmup 5 --no-cpu-baseline > $OUT/avd.log 2>&1
find $OUT/avd -name "*kernel_stats.csv" -exec cp {} $OUT/avatar_default_kernel_stats.csv \;
$TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3d -o c3d -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c3d.log 2>&1
find $OUT/c3d -name "*kernel_stats.csv" -exec cp {} $OUT/cfg3_default_kernel_stats.csv \;
$TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 --eager > $OUT/tr.log 2>&1
find $OUT/tr -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
# one graph-replayed training step as a timeline (which kernels overlap, who waits for whom)
$TO rocprofv3 --kernel-trace --output-format csv -d $OUT/train_trace -o tt -- python3 $ROOT/bench.py --workload train --steps 10 --warmup 3 > $OUT/train_trace.log 2>&1
python3 $ROOT/tools/step_trace.py $(find $OUT/train_trace -name "*kernel_trace.csv" | head -1) > $OUT/train_step_trace.log 2>&1
# counters: separate passes
for ctr in FETCH_SIZE WRITE_SIZE; do
  $TO rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/pmc_$ctr.log 2>&1
  $TO rocprofv3 --pmc $ctr --output-format csv -d $OUT/avatar/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --workload avatar --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/avatar/pmc_$ctr.log 2>&1
done
# the same two counters with 8 frames / cameras per dispatch (per-view traffic of the batched step: tools/pmc_summary.py ... frames_per_launch=8)
for ctr in FETCH_SIZE WRITE_SIZE; do
  $TO rocprofv3 --pmc $ctr --output-format csv -d $OUT/k8/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K8 --no-cpu-baseline > $OUT/k8/pmc_$ctr.log 2>&1
  $TO rocprofv3 --pmc $ctr --output-format csv -d $OUT/avatar_k8/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --workload avatar --steps 5 --warmup 2 $K8 --no-cpu-baseline > $OUT/avatar_k8/pmc_$ctr.log 2>&1
done
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES"
$TO rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_SQ -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/pmc_SQ.log 2>&1
$TO rocprofv3 --pmc $SQ --output-format csv -d $OUT/avatar/pmc_SQ -o c3 -- python3 $ROOT/bench.py --workload avatar --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/avatar/pmc_SQ.log 2>&1
# how busy the vector ALUs are (quad-cycles with a VALU instruction executing, against wave and wait cycles)
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
$TO rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/pmc_SQ2 -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/pmc_SQ2.log 2>&1
$TO rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/avatar/pmc_SQ2 -o c3 -- python3 $ROOT/bench.py --workload avatar --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/avatar/pmc_SQ2.log 2>&1
# the other raster configurations the bench is run on: instruction counts and HBM traffic for their rooflines
#   python tools/pmc_summary.py gpurun_out/prof_r03/cfg2 r03_cfg2 gaussians=50000 width=512 height=512 sh_degree=0
#   python tools/pmc_summary.py gpurun_out/prof_r03/cfg5 r03_cfg5 gaussians=500000 width=2048 height=2048
CFG2="--gaussians 50000 --width 512 --height 512 --sh-degree 0 --forward-only"
CFG5="--gaussians 500000 --width 2048 --height 2048"
for ctr in FETCH_SIZE WRITE_SIZE SQ; do
  set=$ctr; [ $ctr = SQ ] && set="$SQ"
  mkdir -p $OUT/cfg2 $OUT/cfg5
  $TO rocprofv3 --pmc $set --output-format csv -d $OUT/cfg2/pmc_$ctr -o c3 -- python3 $ROOT/bench.py $CFG2 --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/cfg2/pmc_$ctr.log 2>&1
  $TO rocprofv3 --pmc $set --output-format csv -d $OUT/cfg5/pmc_$ctr -o c3 -- python3 $ROOT/bench.py $CFG5 --steps 5 --warmup 2 $K1 --no-cpu-baseline > $OUT/cfg5/pmc_$ctr.log 2>&1
done
# the photometric-loss kernels on their own (1080p noise image, and the avatar frame size with 8 frames per launch): kernel stats + counters
( cd $ROOT && bash tools/prof_loss.sh ${TAG}_loss 1920x1080 > $OUT/loss_prof.log 2>&1; bash tools/pmc_loss.sh ${TAG}_loss 1920x1080 > $OUT/loss_pmc.log 2>&1;
  cp gpurun_out/prof_${TAG}_loss/kernel_stats.csv $OUT/loss_kernel_stats.csv; cp gpurun_out/pmc_${TAG}_loss/summary.csv $OUT/loss_pmc_summary.csv;
  python3 tools/loss_time.py 1920x1080 > $OUT/loss_time.log 2>&1; python3 tools/loss_time.py 512x896 K=8 >> $OUT/loss_time.log 2>&1; python3 tools/loss_time.py 512x896 K=8 avatar >> $OUT/loss_time.log 2>&1 )
cd /tmp
# keep the merge under the 64 MiB limit: drop per-dispatch traces, keep stats + counter files (the three columns of the library's
# own kernels, averaged over the dispatches: what tools/pmc_summary.py reads)
python3 - $OUT <<'PY'
import collections, csv, os, sys
for d, _, fs in os.walk(sys.argv[1]):
    for f in fs:
        if f.endswith("counter_collection.csv"):
            p = os.path.join(d, f)
            acc = collections.OrderedDict()
            rows = [r for r in csv.DictReader(open(p)) if r["Kernel_Name"].startswith(("sg_", "void sg_"))]
            # a run also holds a few dispatches of other shapes (the sizing pass, the one-view-per-step leg of a K-frame run):
            # per kernel only the dispatches with its LARGEST grid count -- the K-frame launches in a K-frame run
            gmax = {}
            for r in rows:
                gmax[r["Kernel_Name"]] = max(gmax.get(r["Kernel_Name"], 0), int(r.get("Grid_Size") or 0))
            for r in rows:
                if int(r.get("Grid_Size") or 0) == gmax[r["Kernel_Name"]]:
                    a = acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
            with open(p, "w", newline="") as fo:          # per (kernel, counter): mean over the dispatches + how many there were
                w = csv.writer(fo); w.writerow(("Kernel_Name", "Counter_Name", "Counter_Value", "Count"))
                w.writerows((k, c, repr(s / n), n) for (k, c), (s, n) in acc.items())
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
