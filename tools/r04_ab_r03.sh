#!/bin/bash
# same-box A/B against the round-3 tree (build/r03: git --work-tree=build/r03 checkout fa44459 -- . ; make): one view at a time
s() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']; print('$1 %8.1f views/s  one view %.4f ms  ' % (j['value'], j['train_step_ms_one_view']), {a[3:-7]: round(b*1e3,1) for a,b in k.items() if b})"; }
for rep in 1 2; do
  (cd build/r03 && python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | s "r03 cfg3")
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | s "now cfg3"
done
(cd build/r03 && python bench.py --workload avatar --steps 60 --warmup 10 --no-cpu-baseline --views-per-step 8 --streams 3 2>/dev/null | s "r03 avatar 8/3")
python bench.py --workload avatar --steps 60 --warmup 10 --no-cpu-baseline --views-per-step 8 --frames-per-launch 1 --streams 3 2>/dev/null | s "now avatar 8/3 K=1"
python bench.py --workload avatar --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | s "now avatar default"
