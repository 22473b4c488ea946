for r in 1 2; do
for L in "" nodirect; do
  if [ -z "$L" ]; then unset SINGS_HIP_LIB; else export SINGS_HIP_LIB=$PWD/sings_amd/libsings_hip_$L.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('${L:-direct}', round(d['value']), d['raster_fwd_bwd_ms_one_view'], 'pp %.1f scan %.1f fwd %.1f' % (1e3*k['sg_preprocess_fwd_kernel'], 1e3*k['sg_tile_scan_kernel'], 1e3*k['sg_render_fwd_kernel']))"
done; done
