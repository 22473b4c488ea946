#!/bin/bash
# On the GPU box: SQ counter passes of the photometric-loss kernels (tools/loss_time.py).  Usage: bash tools/pmc_loss.sh <tag> [loss_time args]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -o c -- python3 $ROOT/tools/loss_time.py "$@" > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv, re, glob
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        if n.startswith("sg_"):
            a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
ks = sorted({k for k, _ in acc})
with open("$OUT/summary.csv", "w") as fo:
    for k in ks:
        for (kk, c), (s, n) in sorted(acc.items()):
            if kk == k:
                line = "%s,%s,%.0f,%d" % (k, c, s / n, n); print(line); fo.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -size +2M -delete
