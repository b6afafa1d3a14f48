#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_api.py -q -x -k "effnet or efficientnet" 2>&1 | tail -5
for rep in 1 2; do
echo -n "unfused, register-tile dw (AVEX_AMD_MBCONV=0 AVEX_AMD_DW_LDS=0): "; AVEX_AMD_MBCONV=0 AVEX_AMD_DW_LDS=0 python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-70
echo -n "shipped: "; python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-70
done
bash scripts/effnet_trace.sh 256 r04_effnet_trace3 > /dev/null 2>&1
AVEX_AMD_MBCONV=0 AVEX_AMD_DW_LDS=0 bash scripts/effnet_trace.sh 256 r04_effnet_trace3u > /dev/null 2>&1
