mkdir -p gpurun_out/r04q; L=$PWD/avex_amd/lib
(
echo "== product (A) vs gemm.hip built with -fno-slp-vectorize (B)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_slp.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04q/ab.txt
cat gpurun_out/r04q/ab.txt
