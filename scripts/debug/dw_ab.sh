#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
for sfx in "" _c1 _c3; do echo -n "lib$sfx: "; AVEX_AMD_LIB=$R/avex_amd/lib/libavexhip$sfx.so python scripts/effnet_bench.py 256 2>&1 | grep -v amdgpu.ids | cut -c1-70; done
done
for sfx in _c1 _c3; do AVEX_AMD_LIB=$R/avex_amd/lib/libavexhip$sfx.so bash scripts/effnet_trace.sh 256 r04_effnet_trace4$sfx > /dev/null 2>&1; echo $sfx; grep -E "mbconv|dwconv_lds" gpurun_out/r04_effnet_trace4$sfx.txt | cut -c1-50,95-; done
