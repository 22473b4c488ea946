#!/bin/bash
# On the GPU box: the round-1 tree (build/r01) against the current tree, same box, same commands.
s() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %9.1f %s  ms/step %.4f' % (j['value'], j['unit'][:8], j['ms_per_step']))"; }
for w in raster avatar; do
  (cd build/r01 && python bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | s "r01  $w batched ")
  python bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | s "now  $w batched "
  (cd build/r01 && python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --views-per-step 1 --streams 1 2>/dev/null | s "r01  $w one view")
  python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --views-per-step 1 --streams 1 2>/dev/null | s "now  $w one view"
done
(cd build/r01 && python bench.py --workload train --steps 30 --warmup 5 2>/dev/null | s "r01  train ")
python bench.py --workload train --steps 30 --warmup 5 2>/dev/null | s "now  train "
