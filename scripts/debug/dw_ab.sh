#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_api.py tests/test_gpu_overflow.py -q -x -k "effnet or efficientnet or skinny or silu" 2>&1 | tail -3
for rep in 1 2 3; do
for sfx in _old ""; do echo -n "lib$sfx: "; AVEX_AMD_LIB=$R/avex_amd/lib/libavexhip$sfx.so python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-70; done
done
