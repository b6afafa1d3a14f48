mkdir -p gpurun_out/r04x; L=$PWD/avex_amd/lib
(
echo "== A: accumulators zeroed by v_mov_b32 per tile (EPI 2) / compiler-peeled K-tile 0 (EPI 1); B: K-tile 0 peeled in the source, C = 0 in its first MFMAs (GEMM_PEEL)"
python scripts/gemm_ab.py --a $L/libavexhip_old.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04x/ab3.txt
cat gpurun_out/r04x/ab3.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_overflow.py tests/test_gpu_api.py -q -x 2>&1 | tail -3
