mkdir -p gpurun_out/r04v; L=$PWD/avex_amd/lib
(
echo "== fc1's bias + GELU epilogue (half output): degree-6 fit of log2 Phi(-a) (A) vs degree-4 (B, -DGEMM_GELU_H=1)"
python scripts/gemm_ab.py --a $L/libavexhip.so --b $L/libavexhip_g4.so --shapes qkv,fc1 --step --rounds 8
AVEX_AMD_LIB=$L/libavexhip.so python scripts/parity_report.py | grep -E "f16.b4.pooled|f16.b4.frame|f16.b4.all_hooks|f16.tone"
AVEX_AMD_LIB=$L/libavexhip_g4.so python scripts/parity_report.py | grep -E "f16.b4.pooled|f16.b4.frame|f16.b4.all_hooks|f16.tone"
) 2>&1 | grep -v amdgpu > gpurun_out/r04v/ab.txt
cat gpurun_out/r04v/ab.txt
