mkdir -p gpurun_out/r04h; L=$PWD/avex_amd/lib
timeout 900 python -m pytest tests/test_gpu_kernels.py -k gemm -x -q 2>&1 | tail -4 > gpurun_out/r04h/tests.txt
(
echo "== prev vs product (slab + side-by-side GELU chains, EPI 2 counted wait, A rows nt for N <= 768)"
python scripts/gemm_ab.py --a $L/libavexhip_prev.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step
echo "== product: A rows nt off (A) vs auto (B)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes out,fc2 --env-a AVEX_AMD_GEMM_A_NT=0 --step
echo "== product: A rows nt auto (A) vs always (B) on the step"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip.so --shapes "" --env-b AVEX_AMD_GEMM_A_NT=1 --step
) 2>&1 | grep -v amdgpu > gpurun_out/r04h/ab.txt
cat gpurun_out/r04h/tests.txt gpurun_out/r04h/ab.txt
