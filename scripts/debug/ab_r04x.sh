mkdir -p gpurun_out/r04x; L=$PWD/avex_amd/lib
(
echo "== A: LDS-DMA addresses as 64-bit pointers per lane; B: scalar base + 32-bit lane offset (GEMM_SADDR)"
python scripts/gemm_ab.py --a $L/libavexhip_old.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04x/ab2.txt
cat gpurun_out/r04x/ab2.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_overflow.py tests/test_gpu_api.py -q -x 2>&1 | tail -3
