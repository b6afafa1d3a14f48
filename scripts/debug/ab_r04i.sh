mkdir -p gpurun_out/r04i; L=$PWD/avex_amd/lib
timeout 900 python -m pytest tests/test_gpu_kernels.py -k gemm -x -q 2>&1 | tail -4 > gpurun_out/r04i/tests.txt
(
echo "== prev vs product (packed row statistics in EPI 2 / EPI 0)"
python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
) 2>&1 | grep -v amdgpu > gpurun_out/r04i/ab.txt
cat gpurun_out/r04i/tests.txt gpurun_out/r04i/ab.txt
