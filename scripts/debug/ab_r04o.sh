mkdir -p gpurun_out/r04o; L=$PWD/avex_amd/lib
(
echo "== this tree (A) vs the same without the EPI 1 epilogue's global stores (B, -DGEMM_NOSTORE=1: arithmetic, LDS transposes kept)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_ns.so --shapes qkv,fc1
echo "== no stores (A) vs no epilogue at all (B)"
python scripts/gemm_ab.py --a $L/libavexhip_ns.so --b $L/libavexhip_ne.so --shapes qkv,fc1
) 2>&1 | grep -v amdgpu > gpurun_out/r04o/ab.txt
cat gpurun_out/r04o/ab.txt
