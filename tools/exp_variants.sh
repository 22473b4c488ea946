#!/bin/bash
# On the GPU box: bench line summaries for the library variants of tools/build_variants.sh.  Usage: bash tools/exp_variants.sh <tag> [bench args] -- <exp numbers>
tag=$1; shift
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
out=gpurun_out/$tag; mkdir -p $out
for n in "$@"; do
  lib=build/exp/libsings_hip_exp$n.so; [ "$n" = "0" ] && lib=sings_amd/libsings_hip.so
  SINGS_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline "${args[@]}" > $out/exp$n.json 2> $out/exp$n.err
  python - <<PY
import json
try:
    j = json.loads(open("$out/exp$n.json").read().strip().splitlines()[-1])
    k = {a.replace("sg_", "").replace("_kernel", ""): round(b * 1e3, 1) for a, b in j["kernel_ms"].items() if b}
    print("exp $n: %.1f views/s  %.4f ms/view batched  %.4f ms one view  " % (j["value"], j["ms_per_view"], j["train_step_ms_one_view"]), k)
except Exception as e:
    print("exp $n failed:", e); print(open("$out/exp$n.err").read()[-800:])
PY
done
