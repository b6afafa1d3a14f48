mkdir -p gpurun_out/r04u; L=$PWD/avex_amd/lib
(
echo "== gemm.hip: v_med3_f32 clamp per stored element (A, -DGEMM_HW_SAT=0) vs MODE.FP16_OVFL saturation (B, product); posconv / effnet kernels saturate in hardware in both"
python scripts/gemm_ab.py --a $L/libavexhip_sat0.so --b $L/libavexhip.so --shapes qkv,out,fc1,fc2 --step --rounds 8
for i in 1 2 3; do python scripts/effnet_bench.py 256 | cut -c1-70; done
) 2>&1 | grep -v amdgpu > gpurun_out/r04u/ab.txt
cat gpurun_out/r04u/ab.txt
python -m pytest tests -q -x -m gpu 2>&1 | tail -3
