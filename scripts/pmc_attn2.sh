#!/bin/bash
# SQ counters of the attention kernel: matrix / vector co-execution and wait reasons (run on the GPU box via gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for grp in "A SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU" \
           "B SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
           "C SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL"; do
  set -- $grp; tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_attn_$tag -- python3 $R/scripts/attn_bench.py 256 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for tag in "ABC":
    for f in glob.glob("$R/gpurun_out/pmc_attn_%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            if "attention2" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        print(tag, {k: round(v / max(n[k], 1) / 1e6, 2) for k, v in acc.items()}, "(millions per launch)")
PY
