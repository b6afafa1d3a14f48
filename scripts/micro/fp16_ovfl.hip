// Does MODE.FP16_OVFL (hwreg MODE bit 23) make the f32 -> f16 conversions of gfx950 saturate at +-65504 instead of producing inf?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp16_ovfl scripts/micro/fp16_ovfl.hip && /tmp/fp16_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, unsigned short* single, unsigned short* packed, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    if (i < n) {
        float a = x[i], b = x[(i + 1) % n];
        _Float16 s;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s) : "v"(a));
        h2 p;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
        single[i] = __builtin_bit_cast(unsigned short, s);
        packed[2 * i] = __builtin_bit_cast(unsigned short, p[0]);
        packed[2 * i + 1] = __builtin_bit_cast(unsigned short, p[1]);
    }
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
    const float hx[] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 1e6f, -1e6f, 3e38f, INFINITY, -INFINITY, NAN, 6e-8f, -70000.0f};
    const int n = sizeof(hx) / sizeof(float);
    float* dx; unsigned short *ds, *dp;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&ds, 2 * n); hipMalloc(&dp, 4 * n);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, n, ds, dp, ovfl);
        unsigned short hs[64], hp[128];
        hipMemcpy(hs, ds, 2 * n, hipMemcpyDeviceToHost); hipMemcpy(hp, dp, 4 * n, hipMemcpyDeviceToHost);
        printf("FP16_OVFL = %d\n", ovfl);
        for (int i = 0; i < n; ++i) printf("  x = %-12g  v_cvt_f16_f32 -> 0x%04x   v_cvt_pk_f16_f32 -> 0x%04x (second lane of pair: 0x%04x)\n", hx[i], hs[i], hp[2 * i], hp[2 * i + 1]);
    }
    return 0;
}
