#!/bin/bash
# avatar workload, per-variant:  bash tools/r05_avatar_variants.sh <variant> ...   (build/exp/lib_<variant>.so; "base" = the tree's library)
ROOT=$(pwd)
for v in base "$@"; do
  if [ $v = base ]; then unset SINGS_HIP_LIB; else export SINGS_HIP_LIB=$ROOT/build/exp/lib_$v.so; fi
  timeout 200 python3 $ROOT/bench.py --workload avatar --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']
print('%-10s frames/s %7.0f  one-frame %.4f ms  bwd composite %.1f us  fwd %.1f us' % ('$v', j['value'], j['train_step_ms_one_view'], 1e3*k['sg_render_bwd_kernel'], 1e3*k['sg_render_fwd_kernel']))"
done
