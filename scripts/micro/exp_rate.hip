// Issue cost of the transcendental instructions on gfx950, one wave per SIMD: cycles per instruction of a stream of INDEPENDENT
// v_exp_f32 / v_exp_f16 / v_rcp_f32 / v_rcp_f16 / v_fma_f32 (s_memtime around 4096 instructions, 8 independent registers).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_rate scripts/micro/exp_rate.hip && /tmp/exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float seed) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; ++it) {
#define E32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define E16(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
#define R32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define R16(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[i]));
#define F32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define L32(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
        if (OP == 0) { REP8(E32) } else if (OP == 1) { REP8(E16) } else if (OP == 2) { REP8(R32) } else if (OP == 3) { REP8(R16) } else if (OP == 4) { REP8(F32) } else { REP8(L32) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)s; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16 * 256);
    const char* names[] = {"v_exp_f32", "v_exp_f16", "v_rcp_f32", "v_rcp_f16", "v_fma_f32", "v_log_f32"};
    for (int op = 0; op < 6; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, 0.5f);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, 0.5f);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, 0.5f);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, 0.5f);
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, 0.5f);
            if (op == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, 0.5f);
        }
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        // s_memtime counts at a fixed 100 MHz on gfx950; report instructions per microsecond of one wave instead of cycles
        printf("%-10s  %6.1f ns per instruction (one wave per SIMD, 4096 independent instructions)\n", names[op], h[0] * 10.0 / 4096.0);
    }
    return 0;
}
