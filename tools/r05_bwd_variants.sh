#!/bin/bash
# per-variant duration of the backward composite, one cfg3 view at a time:  bash tools/r05_bwd_variants.sh <variant> ...
ROOT=$(pwd)
for v in base "$@"; do
  if [ $v = base ]; then unset SINGS_HIP_LIB; else export SINGS_HIP_LIB=$ROOT/build/exp/lib_$v.so; fi
  timeout 120 python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']
print('%-14s views/s %7.0f  one-view %.4f ms  bwd composite %.1f us  fwd %.1f us' % ('$v', j['value'], j['raster_fwd_bwd_ms_one_view'], 1e3*k['sg_render_bwd_kernel'], 1e3*k['sg_render_fwd_kernel']))"
done
