#!/bin/bash
# rocprofv3 kernel durations, one view at a time, round-3 tree (build/r03) vs this tree, same box
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for which in r03 now r03 now; do
  D=$ROOT; [ $which = r03 ] && D=$ROOT/build/r03
  OUT=$ROOT/gpurun_out/abtrace_$which; rm -rf $OUT; mkdir -p $OUT
  (cd $D && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $D/bench.py --steps 60 --warmup 10 --views-per-step 1 --streams 1 --no-cpu-baseline "$@" > $OUT/log 2>&1)
  f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
  python3 - $f $which <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("sg_", "void sg_"))]
print(sys.argv[2], {r["Name"].split("(")[0].replace("void ", "")[:34]: round(float(r["AverageNs"]) / 1e3, 2) for r in rows if float(r["TotalDurationNs"]) > 1e6})
PY
  rm -rf $OUT/t
done
