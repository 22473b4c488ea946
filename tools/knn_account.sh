#!/bin/bash
# Accounting build of the k-NN query (SG_KNN_ACCOUNT: per-point counters) -> build/dbg/libsings_hip_knnacct.so, then
# tools/knn_account.py prints what a point costs.  bash tools/knn_account.sh  (on the GPU box)
set -e
cd "$(dirname "$0")/../sings_amd/csrc"
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
mkdir -p ../../build/dbg
/opt/rocm/bin/hipcc $F -DSG_KNN_ACCOUNT -c sg_reg.hip -o ../../build/dbg/sg_reg_acct.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/dbg/libsings_hip_knnacct.so sg_api.o sg_preprocess.o sg_binning.o sg_render.o sg_skin.o sg_rot.o sg_loss.o ../../build/dbg/sg_reg_acct.o sg_decode.o sg_linear.o
cd ../..
[ "$1" = "build" ] || SINGS_HIP_LIB=$PWD/build/dbg/libsings_hip_knnacct.so python3 tools/knn_account.py
