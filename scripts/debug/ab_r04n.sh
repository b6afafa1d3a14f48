mkdir -p gpurun_out/r04n; L=$PWD/avex_amd/lib
(
echo "== round 3's library (A) vs this tree (B): the four layer GEMMs in the step's forms and the whole 256-clip step, one process"
python scripts/gemm_ab.py --a $L/libavexhip_r03.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
echo "== this tree (A) vs the same with NO epilogue at all (B, -DGEMM_NOEPI=1): the ceiling of any epilogue-hiding scheme"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_ne.so --shapes qkv,out,fc1,fc2
echo "== this tree (A) vs column-walk code compiled in, default order (B)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_cw.so --shapes qkv,fc1
echo "== column-walk library: default order (A) vs groups of 5 for QKV (B)"
python scripts/gemm_ab.py --a $L/libavexhip_cw.so --b $L/libavexhip_cw.so --shapes qkv --env-b AVEX_AMD_GEMM_TILE_ORDER=-5 --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04n/ab.txt
cat gpurun_out/r04n/ab.txt
