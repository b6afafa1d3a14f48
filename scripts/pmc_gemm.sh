cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/gemm_bench.py --shapes qkv,out,fc1,fc2,sq --iters 10 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/scripts/gemm_bench.py --shapes fc2,fc1 --iters 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/scripts/gemm_bench.py --shapes fc2,fc1 --iters 2 > /dev/null 2>&1
ls -R $R/gpurun_out/pmc1 | head; 
