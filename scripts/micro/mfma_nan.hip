// Does v_mfma_f32_16x16x32_f16 / 32x32x16 propagate a NaN or an infinity in an operand -- with and without MODE.FP16_OVFL?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_nan.hip -o scripts/micro/bin/mfma_nan
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, int mode, int ovfl) {
    const int lane = threadIdx.x;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");      // MODE.FP16_OVFL, as the GEMM kernels set it for their f16 conversions
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
    if (lane == 5) b[3] = mode == 0 ? (_Float16)NAN : (mode == 1 ? (_Float16)INFINITY : (_Float16)65504.0f);      // B[k = 8 (5 >> 4) + 3 = 3][col 5]
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
    f32x16 c2; for (int i = 0; i < 16; ++i) c2[i] = 0.f;
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[256 + lane * 16 + i] = c2[i];
}
int main() {
    float* d; float h[256 + 1024];
    hipMalloc(&d, sizeof(h));
    const char* names[3] = {"NaN", "inf", "65504"};
    for (int ovfl = 0; ovfl < 2; ++ovfl)
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode, ovfl);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int n1 = 0, i1 = 0, n2 = 0, i2 = 0;
        for (int i = 0; i < 256; ++i) { n1 += isnan(h[i]); i1 += isinf(h[i]); }
        for (int i = 0; i < 1024; ++i) { n2 += isnan(h[256 + i]); i2 += isinf(h[256 + i]); }
        printf("MODE.FP16_OVFL = %d, operand element = %s: 16x16x32 D has %d NaN %d inf (column 5 = 16 elements); 32x32x16 D has %d NaN %d inf (column 5 = 32 elements)\n", ovfl, names[mode], n1, i1, n2, i2);
    }
    return 0;
}
