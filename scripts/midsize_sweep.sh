python scripts/gemm_midsize.py 2>&1 | tail -9
for mt in 0 64 128 192 256; do
  AVEX_AMD_GEMM_256_MIN_TILES=$mt python scripts/fold_threshold.py auto:1024 auto:4096 auto:8192 auto:16384 2>&1 | grep -v amdgpu.ids | tail -4
done
