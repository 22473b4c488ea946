#!/bin/bash
# Build the library of another revision next to the current one, for same-box A/B runs (box-to-box spread of the bench is ~3 %,
# larger than most single changes):   bash tools/ab_build.sh HEAD~1     ->  build/exp/lib_HEAD~1.so
# then on the GPU box:   SINGS_HIP_LIB=build/exp/lib_HEAD~1.so python bench.py --no-cpu-baseline    vs    python bench.py ...
set -e
cd "$(dirname "$0")/.."
rev=${1:?git revision}
d=build/ab_$(echo "$rev" | tr '/~^' '___')
rm -rf "$d"; mkdir -p "$d" build/exp
git --work-tree="$d" checkout "$rev" -- sings_amd/csrc include
git reset -q
make -C "$d/sings_amd/csrc" -j8 > "$d/make.log" 2>&1 || { tail -20 "$d/make.log"; exit 1; }
cp "$d/sings_amd/libsings_hip.so" "build/exp/lib_$(echo "$rev" | tr '/' '_').so"
echo "built build/exp/lib_$(echo "$rev" | tr '/' '_').so"
