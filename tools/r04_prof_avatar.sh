#!/bin/bash
# kernel trace of the avatar step, K frames per launch on one stream: bash tools/r04_prof_avatar.sh <K> [streams] [views]
K=${1:-8}; S=${2:-1}; V=${3:-8}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04_avatar_K${K}_S${S}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --workload avatar $EXTRA --steps 20 --warmup 5 --views-per-step $V --frames-per-launch $K --streams $S --no-cpu-baseline > $OUT/log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - $f $OUT/timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one steady-state step near the end of the timed region: find the last occurrence of the first kernel of a step
names = [r["Kernel_Name"] for r in rows]
gmax = max(int(r["Grid_Size_X"]) for r in rows if "sg_skin_fwd_kernel" in r["Kernel_Name"])
idx = [i for i, n in enumerate(names) if "sg_skin_fwd_kernel" in n and int(rows[i]["Grid_Size_X"]) == gmax]   # the K-frame launches
out = open(sys.argv[2], "w")
if len(idx) > 12:
    a, b = idx[-6], idx[-5]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}\n")
PY
cat $OUT/timeline.txt | head -60
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
