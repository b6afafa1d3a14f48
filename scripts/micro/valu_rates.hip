// Issue cost in shader cycles per wave64 instruction of the vector instructions the attention softmax can be built from (gfx950),
// one wave per SIMD and two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/valu_rates.hip -o scripts/micro/bin/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
template <int OP>
__global__ __launch_bounds__(512) void rate_kernel(int iters, int waves, unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v0 = 0.5f + lane * 1e-3f, v1 = 0.25f, v2 = 0.125f, v3 = 1.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {0.1f, 0.2f}, p3 = {0.3f, 0.4f};
    float w[8]; f2 q[8];
    for (int i = 0; i < 8; ++i) { w[i] = 0.01f * (lane + i); q[i] = (f2){0.01f * i, 0.02f * lane}; }
    unsigned h0 = 0x3c003c00u, h1 = 0x38003800u, h2 = 0x34003400u, h3 = 0x30003000u;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (waves > 8 && wave >= 4) {          // partner waves: a continuous stream of matrix instructions (waves == 9: 32x32x16, 10: 16x16x32)
        typedef float f16v __attribute__((ext_vector_type(16)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
        f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
        f4v d0 = {}, d1 = {}, d2 = {}, d3 = {};
        for (int it = 0; it < iters * 12; ++it) {
            if (waves == 9) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            } else {
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
            }
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3] == 1234.5f) sink[1] = 1.f;
    }
    if (wave < (waves > 8 ? 4 : waves)) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            // four independent chains per instruction type so that dependent-issue latency does not limit the rate
            if (OP == 0) asm volatile(REP64("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (OP == 1) asm volatile(REP64("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
            if (OP == 2) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (OP == 3) asm volatile(REP64("v_pk_fma_f16 %0, %0, %1, %2\n v_pk_fma_f16 %1, %1, %2, %3\n v_pk_fma_f16 %2, %2, %3, %0\n v_pk_fma_f16 %3, %3, %0, %1\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
            if (OP == 4) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (OP == 5) asm volatile(REP64("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %7\n v_cvt_pk_f16_f32 %3, %7, %4\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
            if (OP == 6) asm volatile(REP64("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (OP == 7) asm volatile(REP64("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(0));
            if (OP == 8) asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            if (OP == 9) asm volatile(REP64("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (OP == 10) asm volatile(REP64("v_exp_legacy_f32 %0, %0\n v_exp_legacy_f32 %1, %1\n v_exp_legacy_f32 %2, %2\n v_exp_legacy_f32 %3, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (OP == 11) asm volatile(REP64("v_pk_mul_f16 %0, %0, %1\n v_pk_mul_f16 %1, %1, %2\n v_pk_mul_f16 %2, %2, %3\n v_pk_mul_f16 %3, %3, %0\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
            if (OP == 12) asm volatile(REP64("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
            if (OP == 13) asm volatile(REP64("v_lshl_add_u32 %0, %1, 23, %0\n v_lshl_add_u32 %1, %2, 23, %1\n v_lshl_add_u32 %2, %3, 23, %2\n v_lshl_add_u32 %3, %0, 23, %3\n") : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
            // sixteen independent chains: the pipe's throughput rather than its latency
            if (OP >= 20) {
#define C16(INS, A) INS " %0, %0" A "\n" INS " %1, %1" A "\n" INS " %2, %2" A "\n" INS " %3, %3" A "\n" INS " %4, %4" A "\n" INS " %5, %5" A "\n" INS " %6, %6" A "\n" INS " %7, %7" A "\n"
                if (OP == 20) asm volatile(REP8(REP8(C16("v_exp_f32", ""))) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
                if (OP == 21) asm volatile(REP8(REP8(C16("v_fma_f32", ", %8, %8"))) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(v1));
                if (OP == 22) asm volatile(REP8(REP8(C16("v_pk_fma_f32", ", %8, %8"))) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(p1));
                if (OP == 23) asm volatile(REP8(REP8(C16("v_add_f32", ", %8"))) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(v1));
                if (OP == 24) asm volatile(REP8(REP8(C16("v_pk_add_f32", ", %8"))) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(p1));
                if (OP == 25) asm volatile(REP8(REP8(C16("v_cvt_pk_f16_f32", ", %8"))) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(v1));
                if (OP == 26) asm volatile(REP8(REP8(C16("v_pk_mul_f32", ", %8"))) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(p1));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    for (int i = 0; i < 8; ++i) v0 += w[i] + q[i][0] + q[i][1];
    float r = v0 + v1 + v2 + v3 + p0[0] + p1[1] + p2[0] + p3[1] + (float)(h0 ^ h1 ^ h2 ^ h3);
    if (r == 1234.5f) sink[0] = r;
    if (lane == 0 && blockIdx.x == 0 && wave < waves) cyc[wave] = t1 - t0;
}
template <int OP> void run(const char* name, unsigned long long* cyc, float* sink) {
    const int iters = 50;
    double res[4];
    for (int w = 0; w < 4; ++w) {
        const int waves = w == 0 ? 4 : w == 1 ? 8 : w == 2 ? 9 : 10;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(512), 0, 0, iters, waves, cyc, sink); CK(hipDeviceSynchronize()); }
        unsigned long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
        res[w] = (double)h[0] / (iters * (OP >= 20 ? 512.0 : 256.0));
    }
    printf("%-22s: %5.1f cycles / instruction alone on the SIMD, %5.1f beside a wave doing the same, %5.1f beside 32x32x16 MFMAs, %5.1f beside 16x16x32 MFMAs\n", name, res[0], res[1], res[2], res[3]);
}
int main() {
    unsigned long long* cyc; float* sink; CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&sink, 64));
    run<0>("v_exp_f32", cyc, sink); run<10>("v_exp_legacy_f32", cyc, sink); run<1>("v_exp_f16", cyc, sink);
    run<2>("v_pk_fma_f32", cyc, sink); run<4>("v_fma_f32", cyc, sink); run<3>("v_pk_fma_f16", cyc, sink); run<11>("v_pk_mul_f16", cyc, sink);
    run<8>("v_pk_add_f32", cyc, sink); run<9>("v_add_f32", cyc, sink); run<5>("v_cvt_pk_f16_f32", cyc, sink); run<6>("v_rndne_f32", cyc, sink);
    run<7>("v_ldexp_f32", cyc, sink); run<12>("v_cvt_i32_f32", cyc, sink); run<13>("v_lshl_add_u32", cyc, sink);
    printf("-- eight independent chains --\n");
    run<20>("v_exp_f32 x8", cyc, sink); run<21>("v_fma_f32 x8", cyc, sink); run<22>("v_pk_fma_f32 x8", cyc, sink); run<23>("v_add_f32 x8", cyc, sink);
    run<24>("v_pk_add_f32 x8", cyc, sink); run<25>("v_cvt_pk_f16_f32 x8", cyc, sink); run<26>("v_pk_mul_f32 x8", cyc, sink);
    return 0;
}
