#!/bin/bash
# round 4: per-record valid bytes instead of zero-filled records -- tests, then the avatar step (bash tools/r04_valid.sh)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04c_valid_tests.log
tail -3 gpurun_out/r04c_valid_tests.log
for cfg in "16 8 2" "8 1 3" "8 8 1"; do
  set -- $cfg
  timeout 600 python bench.py --workload avatar --no-cpu-baseline --steps 20 --warmup 5 --views-per-step $1 --frames-per-launch $2 --streams $3 \
      > gpurun_out/r04c_avatar_V$1_K$2_S$3.json 2> gpurun_out/r04c_avatar_V$1_K$2_S$3.err
  python - "$1" "$2" "$3" <<'PY'
import json,sys
try:
    j=json.loads(open(f'gpurun_out/r04c_avatar_V{sys.argv[1]}_K{sys.argv[2]}_S{sys.argv[3]}.json').read().strip().splitlines()[-1])
    print(f"avatar V={sys.argv[1]} K={sys.argv[2]} streams={sys.argv[3]}: {j['value']:.0f} frames/s, one frame {j['train_step_ms_one_view']*1e3:.1f} us", {k:round(v*1e3,1) for k,v in j['kernel_ms'].items() if v})
except Exception as e: print('failed', sys.argv[1:], e)
PY
done
bash tools/r04_prof_avatar.sh 8 1 8 | tail -40
