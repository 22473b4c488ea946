#!/bin/bash
# forward-composite experiments on the avatar frame (one frame at a time): kernel x tile schedule x wave priorities
run() { python bench.py --workload avatar --steps 40 --warmup 10 --no-cpu-baseline --views-per-step 1 --streams 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']; print('one frame %.3f ms; fwd %.1f us, sort %.1f, scan %.1f, bwd %.1f' % (j['ms_per_step'], k['sg_render_fwd_kernel']*1e3, k['sg_tile_sort_kernel']*1e3, k['sg_tile_scan_kernel']*1e3, k['sg_render_bwd_kernel']*1e3))"; }
for any in 0 1; do for lpt in 0 1; do for prio in 0 1; do
  echo "== SG_FWD_ANY=$any SG_FWD_LPT=$lpt SG_FWD_PRIO=$prio"; SG_FWD_ANY=$any SG_FWD_LPT=$lpt SG_FWD_PRIO=$prio run
done; done; done
for any in 0 1; do
echo "== tile clocks, SG_FWD_ANY=$any (static map, no priorities)"
SG_FWD_ANY=$any SG_FWD_LPT=0 SINGS_HIP_LIB=$PWD/build/dbg/libsings_hip_clk.so python tools/tile_clock.py 2>&1 | grep -v amdgpu.ids
done
echo "== tile clocks, batch kernel + priorities"
SG_FWD_ANY=0 SG_FWD_LPT=0 SG_FWD_PRIO=1 SINGS_HIP_LIB=$PWD/build/dbg/libsings_hip_clk.so python tools/tile_clock.py 2>&1 | grep -v amdgpu.ids
