#!/bin/bash
# On the GPU box: one SQ counter pass over the training-step bench; per-kernel means for kernels matching a pattern.
TAG=$1; PAT=${2:-sg_linear}; shift; shift
CTRS=${@:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o tr -- python3 $ROOT/bench.py --workload train --steps 6 --warmup 2 --eager > $OUT/log.txt 2>&1
python3 - <<PY
import csv, re, glob
from collections import defaultdict
f = glob.glob("$OUT/**/tr_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    if "$PAT" in n:
        a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
ks = sorted({k for k, _ in acc}); cs = sorted({c for _, c in acc})
print("kernel," + ",".join(cs))
for k in ks:
    print(k + "," + ",".join("%.0f" % (acc[(k, c)][0] / max(acc[(k, c)][1], 1)) for c in cs))
PY
