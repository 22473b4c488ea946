#!/bin/bash
# HERE (build container): a second library with direct binning compiled out, next to the real one, for same-box A/B runs:
#   bash tools/ab_direct.sh build     -> sings_amd/libsings_hip_nodirect.so (git-ignored, travels with gpurun)
# On the GPU box:  SINGS_HIP_LIB=$PWD/sings_amd/libsings_hip_nodirect.so python bench.py ...   against   python bench.py ...
set -e
cd "$(dirname "$0")/.."
D=/tmp/sg_nodirect; mkdir -p $D
for f in sg_api sg_preprocess sg_binning sg_render sg_skin sg_rot sg_loss sg_reg sg_decode sg_linear; do
  X=""; [ $f = sg_loss ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -DSG_NO_DIRECT $X -c sings_amd/csrc/$f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sings_amd/libsings_hip_nodirect.so $D/*.o
ls -la sings_amd/libsings_hip_nodirect.so
