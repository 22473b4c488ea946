#!/bin/bash
# per-variant duration of one kernel in the K-frame raster step:  bash tools/r04_ablate_run.sh <kernel substring> <variant> ...
KERN=$1; shift
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  OUT=$ROOT/gpurun_out/abl_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = base ]; then unset SINGS_HIP_LIB; else export SINGS_HIP_LIB=$ROOT/build/exp/lib_$v.so; fi
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --steps 10 --warmup 3 --views-per-step 8 --frames-per-launch 8 --streams 1 --no-cpu-baseline > $OUT/log 2>&1
  python3 - $OUT $KERN $v <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
allrows = list(csv.DictReader(open(f)))
for kern in sys.argv[2].split(","):
    rows = [r for r in allrows if kern in r["Kernel_Name"]]
    if not rows:
        continue
    g = lambda r: int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    gmax = max(g(r) for r in rows)
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if g(r) == gmax)      # the K-frame launches
    print(f"{sys.argv[3]:14s} {kern:28s}: median {d[len(d)//2]:8.1f} us  min {d[0]:8.1f}  n={len(d)}")
import json
try:
    j = json.loads(open(sys.argv[1] + "/log").read().strip().splitlines()[-1]); print(f"{sys.argv[3]:14s} bench: {j['value']:.0f} frames/s under the profiler")
except Exception as e:
    pass
PY
  rm -rf $OUT/t
done
