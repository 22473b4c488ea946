#!/bin/bash
# kernel trace of the pipelined avatar step (two streams by kind of kernel): bash tools/r04_prof_pipe.sh <V> <K> [extra bench args]
V=${1:-16}; K=${2:-8}; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04_pipe; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --workload avatar --steps 10 --warmup 3 --views-per-step $V --frames-per-launch $K --no-cpu-baseline "$@" > $OUT/log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = lambda r: int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)
sk = [i for i, r in enumerate(rows) if "sg_skin_fwd_kernel" in r["Kernel_Name"]]
gmax = max(g(rows[i]) for i in sk)
idx = [i for i in sk if g(rows[i]) == gmax]
# one step = n_batches skin_fwd launches: print ~2 steps from the middle of the timed region
a = idx[len(idx) // 2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:a + 70]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>2}  {r['Kernel_Name'][:46]}")
PY
rm -rf $OUT/t
