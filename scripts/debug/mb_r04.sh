#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x -k "effnet" 2>&1 | tail -15
AVEX_AMD_MBCONV=0 python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids
python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids
bash scripts/effnet_trace.sh 256 r04_effnet_trace2 > /dev/null 2>&1
grep -E "mbconv|launches" gpurun_out/r04_effnet_trace2.txt | tail -12
