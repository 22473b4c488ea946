#!/bin/bash
# round 4: K frames per launch -- tests first, then same-box A/B of the avatar step (python tools/r04_frames.sh on the GPU box)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_frames.py -x -q 2>&1 | tail -15 > gpurun_out/r04b_frames_tests.log
tail -3 gpurun_out/r04b_frames_tests.log
for cfg in "1 3" "8 1" "4 2" "4 1" "2 3" "16 1"; do
  set -- $cfg
  if [ "$1" = "16" ]; then V=16; else V=8; fi
  timeout 600 python bench.py --workload avatar --no-cpu-baseline --steps 20 --warmup 5 --views-per-step $V --frames-per-launch $1 --streams $2 \
      > gpurun_out/r04b_avatar_K$1_S$2.json 2> gpurun_out/r04b_avatar_K$1_S$2.err
  python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(f'gpurun_out/r04b_avatar_K{sys.argv[1]}_S{sys.argv[2]}.json').read().strip().splitlines()[-1])
    print(f"avatar K={sys.argv[1]} streams={sys.argv[2]}: {j['value']:.0f} frames/s, {j['ms_per_view']*1e3:.1f} us/frame, one frame {j['train_step_ms_one_view']*1e3:.1f} us", {k:round(v*1e3,1) for k,v in j['kernel_ms'].items() if v})
except Exception as e: print('failed', sys.argv[1:], e)
PY
done
