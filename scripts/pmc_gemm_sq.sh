#!/bin/bash
# SQ counters of the dominant kernel (gemm256p_kernel), one rocprofv3 --pmc pass set PER GEMM SHAPE, in the forms the 256-clip step launches
# (scripts/gemm_forms.py).  Counters in their own runs with --kernel-trace only; the program goes directly after `--`.
#   bash scripts/pmc_gemm_sq.sh [tag]   ->  gpurun_out/<tag>_gemm_sq.json  (copy to profiles/)
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for sh in qkv out fc1 fc2; do
  for g in A B; do
    rm -rf $R/gpurun_out/${TAG}_sq_${sh}_$g
    if [ $g = A ]; then C="$A"; else C="$B"; fi
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_sq_${sh}_$g -- python3 $R/scripts/gemm_forms.py --shapes $sh --iters 6 > /dev/null 2>&1
  done
done
python3 $R/scripts/parse_gemm_sq.py $R/gpurun_out $TAG
