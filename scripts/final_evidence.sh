#!/bin/bash
# evidence on the final tree of a round (run on the GPU box through gpurun):  bash scripts/final_evidence.sh <tag>
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/${TAG}_gputests.txt
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
python tests/tools/parity_families.py > gpurun_out/${TAG}_parity.log 2>&1
python scripts/parity_report.py > gpurun_out/${TAG}_parity_report.log 2>&1
{ python scripts/other_configs.py; python scripts/eat_bench.py; python scripts/effnet_bench.py 256; python scripts/effnet_bench.py 1024; python scripts/aves_bench.py 128; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_other_configs.txt
python tests/tools/fuzz_kernels.py 300 41 > gpurun_out/${TAG}_fuzz_kernels.txt 2>&1
python tests/tools/fuzz_e2e.py 150 23 > gpurun_out/${TAG}_fuzz_e2e.txt 2>&1
python scripts/soak.py 90 > gpurun_out/${TAG}_soak.txt 2>&1
cat gpurun_out/${TAG}_gputests.txt
tail -qn 2 gpurun_out/${TAG}_fuzz_kernels.txt gpurun_out/${TAG}_fuzz_e2e.txt gpurun_out/${TAG}_soak.txt
cat gpurun_out/${TAG}_other_configs.txt
cat gpurun_out/${TAG}_bench.json
