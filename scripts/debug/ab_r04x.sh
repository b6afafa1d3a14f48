mkdir -p gpurun_out/r04x; L=$PWD/avex_amd/lib
(
echo "== A: before; B: __builtin_assume(nk >= 2) in gemm256p_kernel (the compiler then peels K-tile 0 with C = 0 in the EPI 1 kernels and drops the second zeroing in the EPI 2 ones)"
python scripts/gemm_ab.py --a $L/libavexhip_old.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04x/ab.txt
cat gpurun_out/r04x/ab.txt
