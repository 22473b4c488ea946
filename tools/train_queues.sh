#!/bin/bash
# GPU box: which HIP queue (stream) does what in the training step?  rocprofv3 kernel trace of a few graph replays, per-queue busy time
# and the kernels of the busiest queue.  bash tools/train_queues.sh
ROOT=$(pwd); OUT=$ROOT/gpurun_out/train_queues; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --workload train --steps 8 --warmup 2 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, re
from collections import defaultdict
f = glob.glob("$OUT/t/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 4 replays: find the last 4 occurrences of sg_render_bwd_kernel
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sg_render_bwd_kernel")]
lo = int(rows[idx[-5]]["End_Timestamp"]); hi = int(rows[idx[-1]]["End_Timestamp"])
sel = [r for r in rows if lo < int(r["Start_Timestamp"]) <= hi]
per_q = defaultdict(float); kq = defaultdict(lambda: defaultdict(float))
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    q = r["Queue_Id"]; per_q[q] += d
    kq[q][re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:48]] += d
print(f"4 steps = {(hi - lo) / 1e3:.0f} us wall ({(hi - lo) / 4e3:.0f} us per step)")
for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: busy {t / 4:.0f} us per step")
    for k, d in sorted(kq[q].items(), key=lambda kv: -kv[1])[:14]:
        print(f"      {k:48s} {d / 4:8.1f}")
PY
rm -rf $OUT/t
