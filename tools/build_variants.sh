#!/bin/bash
# Builds experiment variants of the library: build/exp/libsings_hip_exp<N>.so with -DSG_EXP=<N> (see sg_common.h).
# Usage: bash tools/build_variants.sh 1 2 3 ...      then on the GPU box: SINGS_HIP_LIB=build/exp/libsings_hip_exp1.so python bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
for n in "$@"; do
  d=build/exp/obj$n; mkdir -p $d
  for f in sings_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    case $b in sg_preprocess|sg_binning|sg_render|sg_skin|sg_api|sg_linear|sg_decode) /opt/rocm/bin/hipcc $FLAGS -DSG_EXP=$n -c $f -o $d/$b.o & ;;
      *) [ -f sings_amd/csrc/$b.o ] && cp sings_amd/csrc/$b.o $d/$b.o ;; esac
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libsings_hip_exp$n.so $d/*.o
  echo built build/exp/libsings_hip_exp$n.so
done
