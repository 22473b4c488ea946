#!/bin/bash
# On the GPU box: one SQ counter pass of the one-view cfg3 bench; per-kernel means.  Usage: bash tools/pmc_k1.sh <tag> [counters...]
TAG=$1; shift
CTRS=${@:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 --views-per-step 1 --streams 1 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, re, glob
from collections import defaultdict
f = glob.glob("$OUT/**/c3_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    if n.startswith("sg_"):
        a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
ks = sorted({k for k, _ in acc})
cs = sorted({c for _, c in acc})
print("kernel," + ",".join(cs))
for k in ks:
    print(k + "," + ",".join("%.0f" % (acc[(k, c)][0] / max(acc[(k, c)][1], 1)) for c in cs))
PY
