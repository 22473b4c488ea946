#!/bin/bash
# avatar step: frames per launch x streams x views per step (same box)
for cfg in "$@"; do
  IFS=, read V K S <<< "$cfg"
  timeout 150 python bench.py --workload avatar --no-cpu-baseline --steps 20 --warmup 5 --views-per-step $V --frames-per-launch $K --streams $S \
      > gpurun_out/r04c_avatar_V${V}_K${K}_S${S}.json 2> gpurun_out/r04c_avatar_V${V}_K${K}_S${S}.err
  python - $V $K $S <<'PY'
import json,sys
V,K,S=sys.argv[1:]
try:
    j=json.loads(open(f'gpurun_out/r04c_avatar_V{V}_K{K}_S{S}.json').read().strip().splitlines()[-1])
    print(f"avatar views/step={V} K={K} streams={S}: {j['value']:.0f} frames/s, {j['ms_per_view']*1e3:.1f} us/frame, one frame {j['train_step_ms_one_view']*1e3:.1f} us")
except Exception as e: print('failed', V,K,S, e, open(f'gpurun_out/r04c_avatar_V{V}_K{K}_S{S}.err').read()[-600:])
PY
done
