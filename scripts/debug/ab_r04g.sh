mkdir -p gpurun_out/r04g; L=$PWD/avex_amd/lib
timeout 600 python -m pytest tests/test_gpu_kernels.py -k gemm -x -q 2>&1 | tail -4 > gpurun_out/r04g/tests.txt
(
echo "== prev vs v1 (EPI1 swap)"; python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip_v1.so --shapes qkv,fc1
echo "== prev vs product (EPI1 + EPI2 swap)"; python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip.so --shapes out,fc2 --step
echo "== prev vs v4 (EPI2 swap only)"; python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip_v4.so --shapes "" --step
echo "== product: default walk vs column groups of 6 (fc1), 5 (qkv), 3 (out)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes fc1 --env-b AVEX_AMD_GEMM_TILE_ORDER=-6
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes fc1 --env-b AVEX_AMD_GEMM_TILE_ORDER=-4
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes qkv --env-b AVEX_AMD_GEMM_TILE_ORDER=-5
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes qkv --env-b AVEX_AMD_GEMM_TILE_ORDER=-3
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes out --env-b AVEX_AMD_GEMM_TILE_ORDER=-3
echo "== A stream nt (v5): default walk, column groups"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_v5.so --shapes qkv,fc1,out
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_v5.so --shapes fc1 --env-b AVEX_AMD_GEMM_TILE_ORDER=-6
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_v5.so --shapes qkv --env-b AVEX_AMD_GEMM_TILE_ORDER=-5
echo "== W stream nt (v6): default walk"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_v6.so --shapes qkv,fc1,out
) 2>&1 | grep -v amdgpu > gpurun_out/r04g/ab.txt
cat gpurun_out/r04g/tests.txt gpurun_out/r04g/ab.txt
