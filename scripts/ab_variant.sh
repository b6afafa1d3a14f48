#!/bin/bash
# Build an A/B variant of libavexhip.so and (optionally) run a command against it -- the one parametrised form of the per-experiment
# scripts earlier rounds kept under scripts/debug/ (ab_r04g.sh ... ab_r04x.sh each named variant libraries nothing in the tree built).
#
#   scripts/ab_variant.sh <suffix> "<extra cflags>" [-- command ...]
#
#   scripts/ab_variant.sh sat0 "-DGEMM_HW_SAT=0"                      # builds avex_amd/lib/libavexhip_sat0.so (objects in avex_amd/_build_sat0/)
#   scripts/ab_variant.sh sat0 "-DGEMM_HW_SAT=0" -- python scripts/gemm_ab.py --a avex_amd/lib/libavexhip_sat0.so --b avex_amd/lib/libavexhip.so --shapes qkv,fc1 --step
#   scripts/ab_variant.sh diag "-DATT_STAMPS=1" -- env AVEX_AMD_LIB=$PWD/avex_amd/lib/libavexhip_diag.so python scripts/attn_stamps.py
#
# The switches the kernels read at compile time (all default to the product form):
#   gemm.hip       GEMM_HW_SAT=0  GEMM_PEEL=0  GEMM_EPI1_SWAP=1  GEMM_GELU_CHAINS=1  GEMM_EPI2_EARLY=1  GEMM_COL_WALK=1  GEMM_SADDR=1  GEMM_NOEPI=1  GEMM_NOSTORE=1
#   attention.hip  ATT_IL=0  ATT_PRIO=0  ATT_STAGGER=n  ATT_BIAS_REUSE=0  ATT_STAMPS=1 (with AVEX_AMD_DIAG=1)
#   attention16.hip ATT3_PRIO=0|2  ATT3_STAGGER=n  ATT3_WG_STAGGER=n  ATT3_FULL_LINES=1  (knock-outs / stamps: scripts/micro/att16_bench.hip, no library needed)
#   melspec.hip    STFT_KNOCK=bits
# Run on the GPU box through gpurun: build here (hipcc cross-compiles), the .so travels with the snapshot.
set -e
[ $# -ge 2 ] || { sed -n 2,20p "$0"; exit 2; }
suffix=$1; cflags=$2; shift 2
cd "$(dirname "$0")/.."
AVEX_AMD_LIB_SUFFIX=$suffix AVEX_AMD_EXTRA_CFLAGS="$cflags" python -m avex_amd.build
echo "[ab_variant] avex_amd/lib/libavexhip_${suffix}.so  (select with AVEX_AMD_LIB=\$PWD/avex_amd/lib/libavexhip_${suffix}.so)"
if [ "$1" = "--" ]; then shift; exec "$@"; fi
