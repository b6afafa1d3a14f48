#!/bin/bash
# round-3 evidence on the final tree (run on the GPU box through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/collect_profiles.sh r03u > gpurun_out/r03u_collect.log 2>&1
python tests/tools/parity_families.py > gpurun_out/r03u_parity.log 2>&1
{ python scripts/other_configs.py; python scripts/eat_bench.py; python scripts/effnet_bench.py 256; python scripts/effnet_bench.py 1024; python scripts/aves_bench.py 128; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r03u_other_configs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03u_effnet_prof -- python3 $R/scripts/effnet_bench.py 256 > /dev/null 2>&1
cp $(find $R/gpurun_out/r03u_effnet_prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r03u_effnet_kernel_stats.csv
cd $R
python tests/tools/fuzz_kernels.py 200 31 > gpurun_out/r03u_fuzz_kernels.txt 2>&1
python tests/tools/fuzz_e2e.py 100 17 > gpurun_out/r03u_fuzz_e2e.txt 2>&1
python scripts/soak.py 60 > gpurun_out/r03u_soak.txt 2>&1
tail -qn 2 gpurun_out/r03u_fuzz_kernels.txt gpurun_out/r03u_fuzz_e2e.txt gpurun_out/r03u_soak.txt
cat gpurun_out/r03u_other_configs.txt
