#!/bin/bash
# in the build container, after `gpurun -- 'bash tools/collect_profiles.sh <tag>'`, in the tree the passes ran from:
#   bash tools/keep_profiles.sh <tag>     -> PMC summaries + sidecars and the campaign's files copied into profiles/<tag>_*
TAG=${1:-r06}; P=gpurun_out/prof_$TAG
AV="workload=avatar gaussians=150000 width=512 height=896 sh_degree=0"
python tools/pmc_summary.py $P $TAG | tail -1 | cut -c1-100
python tools/pmc_summary.py $P/avatar ${TAG}_avatar $AV | tail -1 | cut -c1-100
python tools/pmc_summary.py $P/k8 ${TAG}_k8 frames_per_launch=8 | tail -1 | cut -c1-100
python tools/pmc_summary.py $P/avatar_k8 ${TAG}_avatar_k8 $AV frames_per_launch=8 | tail -1 | cut -c1-100
python tools/pmc_summary.py $P/cfg2 ${TAG}_cfg2 gaussians=50000 width=512 height=512 sh_degree=0 | tail -1 | cut -c1-100
python tools/pmc_summary.py $P/cfg5 ${TAG}_cfg5 gaussians=500000 width=2048 height=2048 | tail -1 | cut -c1-100
for f in bench_default bench_cfg3_k1 bench_avatar bench_avatar_k1 bench_train bench_train_k16 bench_cfg2 bench_cfg5 bench_rccl_world1 bench_rccl_world1_rs_ag; do cp $P/$f.json profiles/${TAG}_$f.json; done
for f in cfg3_k1_kernel_stats cfg3_k1_timeline cfg3_k8_kernel_stats cfg3_k8_timeline cfg3_default_kernel_stats avatar_k1_kernel_stats avatar_k1_timeline avatar_k8_kernel_stats avatar_k8_timeline avatar_default_kernel_stats train_kernel_stats; do cp $P/$f.csv profiles/${TAG}_$f.csv; done
cp $P/train_step_trace.log profiles/${TAG}_train_step_trace.log; cp $P/wrapper_time.log profiles/${TAG}_wrapper_time.log
cp $P/loss_kernel_stats.csv profiles/${TAG}_loss_kernel_stats.csv; cp $P/loss_pmc_summary.csv profiles/${TAG}_loss_pmc.csv; cp $P/loss_time.log profiles/${TAG}_loss_time.log
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
json.dump({"config": {"workload": "loss", "width": 1920, "height": 1080}, "sources": bench.source_hashes(),
           "command": "tools/pmc_loss.sh: rocprofv3 --pmc <set> -- python3 tools/loss_time.py 1920x1080 (four passes)"}, open("profiles/${TAG}_loss_pmc.meta.json", "w"), indent=1)
PY
