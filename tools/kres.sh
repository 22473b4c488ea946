#!/bin/bash
# Register / LDS / scratch use of every kernel of one .hip file (device-only assembly): bash tools/kres.sh sings_amd/csrc/sg_render.hip
F=$1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -S --cuda-device-only -o /tmp/kres.s $F 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|sgpr_count|agpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):" /tmp/kres.s | sed 's/^ *//' | paste - - - - - - - | sed 's/\.//g'
