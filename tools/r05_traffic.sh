#!/bin/bash
# cfg3, 8 cameras per launch on one stream: bench value + the two HBM counters per kernel (separate passes)  -- bash tools/r05_traffic.sh <tag>
TAG=${1:-x}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/traffic_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K8="--views-per-step 8 --frames-per-launch 8 --streams 1"
timeout 200 python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$ctr -o c3 -- python3 $ROOT/bench.py --steps 5 --warmup 2 $K8 --no-cpu-baseline --no-secondary > $OUT/pmc_$ctr.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
j = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("views/s %.0f  ms/view %.4f  one-view %.4f" % (j["value"], j["ms_per_view"], j["raster_fwd_bwd_ms_one_view"]), {k: round(1e3 * v, 1) for k, v in j["kernel_ms"].items() if v})
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ctr, mul, idx in (("FETCH_SIZE", 2.0, 0), ("WRITE_SIZE", 1.0, 1)):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    big = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == ctr:
            big[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in big.items():
        v = sorted(v)[len(v) // 2:]                      # the K-frame launches (the one-view leg's launches are the smaller half)
        tot[k][idx] = mul * 1024 * sum(v) / len(v) / 8  # KiB -> bytes, per view of the 8-camera launch (FETCH_SIZE x 2: gfx950)
for k, (f, w, _) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    if f + w > 1e5:
        print("%-50s fetch %7.1f MB  write %7.1f MB  total %7.1f MB per view" % (k[:50], f / 1e6, w / 1e6, (f + w) / 1e6))
print("sum %.1f MB per view" % (sum(f + w for f, w, _ in tot.values()) / 1e6))
PY
