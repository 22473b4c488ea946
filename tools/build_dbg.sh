#!/bin/bash
# Measurement build of the library with extra -D flags (never the product): bash tools/build_dbg.sh <name> -DSG_TILE_CLOCK ...
# -> build/dbg/libsings_hip_<name>.so; run with SINGS_HIP_LIB=$PWD/build/dbg/libsings_hip_<name>.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); D=$ROOT/build/dbg/$NAME; mkdir -p $D
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
for f in $ROOT/sings_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  case $b in sg_render|sg_binning|sg_preprocess|sg_skin|sg_api) /opt/rocm/bin/hipcc $FLAGS "$@" -c $f -o $D/$b.o & ;; *) cp $ROOT/sings_amd/csrc/$b.o $D/$b.o ;; esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/dbg/libsings_hip_$NAME.so $D/*.o
echo built $ROOT/build/dbg/libsings_hip_$NAME.so
