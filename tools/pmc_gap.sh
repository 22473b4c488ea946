#!/bin/bash
# On the GPU box: the composite kernels with ONE view per launch against 8 cameras per launch -- the same counters for both forms
# (143 vs 128 us backward, 68 vs 62 forward per view).  Usage: bash tools/pmc_gap.sh <tag>
TAG=${1:-gap}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K1="--views-per-step 1 --streams 1"; K8="--views-per-step 8 --frames-per-launch 8 --streams 1"
i=0
for CTRS in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
            "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/k1_$i -o c -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K1 --no-cpu-baseline --no-secondary > $OUT/k1_$i.log 2>&1
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/k8_$i -o c -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K8 --no-cpu-baseline --no-secondary > $OUT/k8_$i.log 2>&1
done
python3 - <<PY
import csv, re, glob
from collections import defaultdict
for form, per in (("k1", 1), ("k8", 8)):
    acc = defaultdict(lambda: [0.0, 0]); gmax = {}
    rows = []
    for f in glob.glob("$OUT/%s_*/**/c_counter_collection.csv" % form, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "sg_render" in r["Kernel_Name"]]
    for r in rows:
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        gmax[n] = max(gmax.get(n, 0), int(r.get("Grid_Size") or 0))
    for r in rows:
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        if int(r.get("Grid_Size") or 0) == gmax[n]:
            a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        print("%s,%s,%s,%.0f,per_view" % (form, k, c, s / n / per))
PY
find $OUT -name "*counter_collection.csv" -size +1M -delete
