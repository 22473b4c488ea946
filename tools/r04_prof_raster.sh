#!/bin/bash
# kernel trace of the cfg3 step, K cameras per launch: bash tools/r04_prof_raster.sh <K> [streams] [views]
K=${1:-8}; S=${2:-1}; V=${3:-8}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_r04_raster_K${K}_S${S}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --steps 10 --warmup 3 --views-per-step $V --frames-per-launch $K --streams $S --no-cpu-baseline > $OUT/log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - $f $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = lambda r: int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)
pre = [i for i, r in enumerate(rows) if "sg_preprocess_fwd_kernel" in r["Kernel_Name"]]
gmax = max(g(rows[i]) for i in pre)
idx = [i for i in pre if g(rows[i]) == gmax]
out = open(sys.argv[2], "w")
a, b = idx[-6], idx[-5]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  {r['Kernel_Name'][:60]}  grid {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}\n")
PY
cat $OUT/timeline.txt | head -30
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
