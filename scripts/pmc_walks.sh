#!/bin/bash
# L2-miss bytes (FETCH_SIZE doubled: the gfx950 rule) per launch of QKV / fc1 / out_proj in the step's forms, default tile walk against the
# column-group walk and the nt policies (variant libraries built with -DGEMM_COL_WALK=1 / -DGEMM_A_POLICY=2).  One rocprofv3 pass each.
#   bash scripts/pmc_walks.sh   (GPU box)  -> gpurun_out/r04_walks.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
L=$R/avex_amd/lib
one() {   # name shape lib tile_order
  rm -rf $R/gpurun_out/pmcw
  AVEX_AMD_LIB=$3 AVEX_AMD_GEMM_TILE_ORDER=$4 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcw -- python3 $R/scripts/gemm_forms.py --shapes $2 --iters 4 > /dev/null 2>&1
  python3 - "$R/gpurun_out/pmcw" "$1" $2 <<'PY'
import csv, glob, sys, os
v = []; d = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]) * 2048.0)
            if r.get("End_Timestamp"): d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
alg = {"qkv": 195 + 3.5, "out": 195 + 195 + 1.2, "fc1": 195 + 4.7, "fc2": 780 + 195 + 4.7}[sys.argv[3]]
v = v[2:] or v
print(f"{sys.argv[2]:34s} {sys.argv[3]:4s}: fetched {sum(v) / max(1, len(v)) / 1e6:8.1f} MB per launch (inputs once: {alg:.0f} MB)" + (f"   {sum(d[2:]) / max(1, len(d[2:])):.0f} us under the profiler" if d else ""))
PY
}
{
one "default walk (8 row panels)"      qkv $L/libavexhip.so 0
one "column groups of 5"               qkv $L/libavexhip_cw.so -5
one "column groups of 3"               qkv $L/libavexhip_cw.so -3
one "default walk (8 row panels)"      fc1 $L/libavexhip.so 0
one "column groups of 6"               fc1 $L/libavexhip_cw.so -6
one "column groups of 4"               fc1 $L/libavexhip_cw.so -4
one "default walk"                     out $L/libavexhip.so 0
one "A rows nt (default walk)"         out $L/libavexhip_v5.so 0
one "A rows nt (default walk)"         qkv $L/libavexhip_v5.so 0
one "A rows nt (default walk)"         fc1 $L/libavexhip_v5.so 0
} | tee $R/gpurun_out/r04_walks.txt
