#!/bin/bash
# SQ counters of the two streamed attention kernels (variant 2: 32x32x16 MFMAs, variant 3: 16x16x32) on the same launch (256 clips x 12 heads x
# 496 tokens, random data): matrix / vector pipe busy, co-execution, wait reasons.  Run on the GPU box via gpurun; writes
# gpurun_out/r05_attention_sq.json (copy to profiles/).  Separate --pmc passes, program directly after `--`.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for v in 2 3; do
  export AVEX_AMD_ATT_VARIANT=$v
  for grp in "A SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU" \
             "B SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" \
             "C SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
    set -- $grp; tag=$1; shift
    rm -rf $R/gpurun_out/pmc_attn_v${v}_$tag
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_attn_v${v}_$tag -- python3 $R/scripts/attn_bench.py 256 4 > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, collections, json
out = {"launch": "256 clips x 12 heads x 496 tokens, f16, random q/k/v, gate + bias (scripts/attn_bench.py 256 4)",
       "units": "millions per launch; SQ_VALU_MFMA_BUSY_CYCLES in cycles, the SQ_ACTIVE_/SQ_WAIT_/SQ_WAVE_/COEXEC counters in quad-cycles (guide, cycle-constants table)",
       "variants": {}}
for v in (2, 3):
    acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for tag in "ABC":
        for f in glob.glob("$R/gpurun_out/pmc_attn_v%d_%s/**/*counter_collection.csv" % (v, tag), recursive=True):
            for row in csv.DictReader(open(f)):
                if ("attention%d_kernel" % v) in row["Kernel_Name"]:
                    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        for f in glob.glob("$R/gpurun_out/pmc_attn_v%d_%s/**/*kernel_trace.csv" % (v, tag), recursive=True):
            for row in csv.DictReader(open(f)):
                if ("attention%d_kernel" % v) in row["Kernel_Name"]:
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    c = {k: acc[k] / max(n[k], 1) / 1e6 for k in acc}
    simd_cycles = 4.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0)            # 4 SIMDs per CU (millions of cycles, summed over CUs)
    d = {"counters_millions": {k: round(x, 2) for k, x in sorted(c.items())}, "us_per_launch_under_profiler": round(sorted(dur)[len(dur) // 2], 1) if dur else None}
    if simd_cycles > 0:
        d["matrix_pipe_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles, 3)
        d["vector_pipe_active_frac"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles, 3)
        d["mfma_valu_coexec_frac"] = round(4.0 * c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / simd_cycles, 3)
        d["valu_per_mfma_instruction"] = round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 1e-9), 1e-9), 2)
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        d["wave_time_split"] = {"waiting_any": round(c.get("SQ_WAIT_ANY", 0) / w, 3), "issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)}
    out["variants"]["attention%d_kernel" % v] = d
json.dump(out, open("$R/gpurun_out/r05_attention_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
