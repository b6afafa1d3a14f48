mkdir -p gpurun_out/r04j; L=$PWD/avex_amd/lib
(
echo "== prev vs product (hot path as prev; packed row statistics in EPI 2 / EPI 0)"
python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04j/ab.txt
cat gpurun_out/r04j/ab.txt
