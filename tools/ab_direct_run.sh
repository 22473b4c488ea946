#!/bin/bash
# On the GPU box, after `bash tools/ab_direct.sh build` in the build container: bench lines with direct binning and with the plain path
# (the same tree compiled with -DSG_NO_DIRECT), A / B / A / B on one box.  Usage: bash tools/ab_direct_run.sh [raster|avatar]
W=${1:-raster}
for r in 1 2; do
for L in "" nodirect; do
  if [ -z "$L" ]; then unset SINGS_HIP_LIB; else export SINGS_HIP_LIB=$PWD/sings_amd/libsings_hip_$L.so; fi
  python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms',{})
print('${L:-direct}', round(d['value']), d.get('raster_fwd_bwd_ms_one_view', d.get('ms_per_frame_one_per_step')), {a:round(1e3*b,1) for a,b in k.items() if b})"
done; done
