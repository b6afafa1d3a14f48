#!/bin/bash
# Round 6, "keep W in L2 for the K = 768 products": the column-group walk with per-XCD contiguous tile ranges (each XCD owns an N range: fc1 two
# halves of 6 column tiles over four XCDs each, QKV groups of 5 / 3) WITH the A stream non-temporal (so that A lines do not age the XCD's W
# slice out of its L2) -- the combination round 4 did not build -- beside the default walk, the column walk alone and A nt alone.
# Per candidate: L2-miss bytes per launch (rocprofv3 --pmc FETCH_SIZE, doubled: the gfx950 rule), then un-profiled time, board watts, clock and
# joules per launch (scripts/gemm_walk_power.py).  Libraries: AVEX_AMD_LIB_SUFFIX=cw   AVEX_AMD_EXTRA_CFLAGS="-DGEMM_COL_WALK=1"
#                                                              AVEX_AMD_LIB_SUFFIX=cwnt AVEX_AMD_EXTRA_CFLAGS="-DGEMM_COL_WALK=1 -DGEMM_A_POLICY=2"
#                                                              AVEX_AMD_LIB_SUFFIX=ant  AVEX_AMD_EXTRA_CFLAGS="-DGEMM_A_POLICY=2"
#   bash scripts/walks_r06.sh   (GPU box)  -> gpurun_out/r06c_walks.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
L=$R/avex_amd/lib
one() {   # label shape lib tile_order
  rm -rf $R/gpurun_out/pmcw
  AVEX_AMD_LIB=$3 AVEX_AMD_GEMM_TILE_ORDER=$4 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcw -- python3 $R/scripts/gemm_forms.py --shapes $2 --iters 4 > /dev/null 2>&1
  python3 - "$R/gpurun_out/pmcw" "$1" $2 <<'PY'
import csv, glob, sys, os
v = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" in r["Kernel_Name"]:
            v.append(float(r["Counter_Value"]) * 2048.0)
alg = {"qkv": 195 + 3.5, "out": 195 + 195 + 1.2, "fc1": 195 + 4.7, "fc2": 780 + 195 + 4.7}[sys.argv[3]]
v = v[2:] or v
f = sum(v) / max(1, len(v)) / 1e6
print(f"{sys.argv[3]:4s} {sys.argv[2]:44s} fetched {f:7.1f} MB per launch = {f / alg:4.2f} x the inputs ({alg:.0f} MB)")
PY
  AVEX_AMD_LIB=$3 AVEX_AMD_GEMM_TILE_ORDER=$4 python3 $R/scripts/gemm_walk_power.py $2 "$1" 2>&1 | grep -v amdgpu.ids
}
{
for sh in qkv fc1; do
  if [ $sh = qkv ]; then G="-5 -3"; else G="-6 -4"; fi
  one "default walk (8 row panels per group)" $sh $L/libavexhip.so 0
  one "A stream nt, default walk" $sh $L/libavexhip_ant.so 0
  for g in $G; do
    one "column groups of ${g#-} per XCD range" $sh $L/libavexhip_cw.so $g
    one "column groups of ${g#-} + A stream nt" $sh $L/libavexhip_cwnt.so $g
  done
done
} | tee $R/gpurun_out/r06c_walks.txt
