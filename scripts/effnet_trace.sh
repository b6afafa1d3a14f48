#!/bin/bash
# Per-launch trace of one EfficientNet-B0 forward (run on the GPU box):  bash scripts/effnet_trace.sh [B] [tag]
B=${1:-256}
TAG=${2:-effnet_trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG} -- python3 $R/scripts/effnet_bench.py $B > $R/gpurun_out/${TAG}.log 2>&1
python3 $R/scripts/parse_effnet_trace.py $(ls -t $(find $R/gpurun_out/${TAG} -name "*kernel_trace.csv") | head -1) | tee $R/gpurun_out/${TAG}.txt
