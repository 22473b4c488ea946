#!/bin/bash
# On the GPU box: the raster headline under other schedules (views per step / frames per launch / streams), same box.  LAB 6.7
for c in "16 8 2" "24 8 3" "32 16 2" "16 16 1" "32 8 4" "48 16 3" "16 8 2"; do
  set -- $c
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --views-per-step $1 --frames-per-launch $2 --streams $3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['value']), round(d['ms_per_view']*1e3,1))"
done
