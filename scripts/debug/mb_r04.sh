#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_api.py -q -x -k "effnet or efficientnet or skinny" 2>&1 | tail -15
AVEX_AMD_MBCONV=0 python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids
python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids
python scripts/effnet_bench.py 1024 2>&1 | grep -v amdgpu.ids
bash scripts/effnet_trace.sh 256 r04_effnet_trace2 > /dev/null 2>&1
tail -120 gpurun_out/r04_effnet_trace2.txt | cut -c1-45,62-200
