#!/bin/bash
# On the GPU box: batched-mode sweep of the cfg3 bench (views per step x streams).
run() { python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.1f views/s  %.4f ms/view  one-view %.4f' % ('$*', j['value'], j['ms_per_view'], j['train_step_ms_one_view']))"; }
run
run --streams 2
run --streams 4
run --views-per-step 12 --streams 3
run --views-per-step 12 --streams 4
run --views-per-step 16 --streams 4
run --views-per-step 4 --streams 2
run --one-shot-reduce
