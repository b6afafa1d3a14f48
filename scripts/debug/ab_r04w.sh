mkdir -p gpurun_out/r04w; L=$PWD/avex_amd/lib
(
echo "== the tree of commit 'fc1 epilogue: degree-4 GELU' (A) vs + relu-form GELU tail, packed LayerNorm-fold FMAs, DPP operand inside the statistics adds (B)"
python scripts/gemm_ab.py --a $L/libavexhip_old.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04w/ab.txt
cat gpurun_out/r04w/ab.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_overflow.py -q -x 2>&1 | tail -3
