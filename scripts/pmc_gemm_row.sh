#!/bin/bash
# SQ counters of the attention output projection on the two kernels: gemm256p_kernel (variant 5) and gemm_row_kernel (variant 8), and fc2's shape
# for the loop alone.  Counters in their own runs with --kernel-trace only; the program goes directly after `--`.
#   bash scripts/pmc_gemm_row.sh [tag]   ->  gpurun_out/<tag>_row_sq_*/  (summarised by scripts/parse_row_sq.py)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for v in 5 8; do
  for sh in out fc2; do
    for g in A B; do
      rm -rf $R/gpurun_out/${TAG}_row_sq_v${v}_${sh}_$g
      if [ $g = A ]; then C="$A"; else C="$B"; fi
      rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_row_sq_v${v}_${sh}_$g -- python3 $R/scripts/gemm_forms.py --shapes $sh --iters 6 --variant $v > /dev/null 2>&1
    done
  done
done
python3 $R/scripts/parse_row_sq.py $R/gpurun_out $TAG
