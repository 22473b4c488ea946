#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the complete training step (bench.py --workload train).  Usage: bash tools/prof_train.sh <tag>
TAG=${1:-tr}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python3 $ROOT/bench.py --workload train --steps 30 --warmup 5 > $OUT/tr.log 2>&1
tail -1 $OUT/tr.log | cut -c1-400
python3 - <<PY
import csv, re
rows = []
for r in csv.DictReader(open("$OUT/tr/tr_kernel_stats.csv")):
    rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), float(r["AverageNs"]), re.sub(r"^void ", "", r["Name"]).split("(")[0][:70]))
rows.sort(reverse=True)
steps = 35 + 15 + 3      # timed + warm-up + sizing/capture passes (approximate: per-step = total / calls-of-a-once-per-step kernel)
once = next(c for t, c, a, n in rows if n.startswith("sg_render_bwd_kernel"))
print("calls of sg_render_bwd_kernel (= steps executed):", once)
for t, c, a, n in rows[:28]:
    print(f"{n:70s} {c:6d} calls  {a / 1e3:9.1f} us avg  {t / once / 1e3:9.1f} us per step")
print("sum per step (us):", sum(t for t, c, a, n in rows) / once / 1e3)
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
