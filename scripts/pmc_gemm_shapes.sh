#!/bin/bash
# L2-miss bytes (FETCH_SIZE, doubled per the gfx950 rule) per launch of each of the four layer GEMMs, one rocprofv3 pass per shape
# (the kernel name is the same for all of them).  Usage (GPU box): bash scripts/pmc_gemm_shapes.sh [extra env assignments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for sh in qkv out fc1 fc2; do
  rm -rf $R/gpurun_out/pmcs_$sh
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcs_$sh -- python3 $R/scripts/gemm_bench.py --shapes $sh --iters 4 > /dev/null 2>&1
  python3 - "$R/gpurun_out/pmcs_$sh" $sh <<'PY'
import csv, glob, sys, os
v = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]) * 2048.0)
ideal = {"qkv": 195 + 3.5, "out": 195 + 195 + 1.2, "fc1": 195 + 4.7, "fc2": 780 + 195 + 4.7}[sys.argv[2]]
print(f"{sys.argv[2]:4s}: {len(v)} launches, fetched {sum(v) / max(1, len(v)) / 1e6:8.1f} MB per launch (operands + residual once: {ideal:.0f} MB)")
PY
done
