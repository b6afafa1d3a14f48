#!/bin/bash
# last evidence pass of round 4: everything final_evidence.sh collects + fresh SQ counters of the GEMM shapes + the EfficientNet fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/final_evidence.sh r04z > gpurun_out/r04z_final.log 2>&1
bash scripts/pmc_gemm_sq.sh r04z > gpurun_out/r04z_gemm_sq.txt 2>&1
python tests/tools/fuzz_effnet.py 40 7 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r04z_fuzz_effnet.txt
cat gpurun_out/r04z_gputests.txt gpurun_out/r04z_gemm_sq.txt gpurun_out/r04z_fuzz_effnet.txt; tail -3 gpurun_out/r04z_final.log | cut -c1-400
