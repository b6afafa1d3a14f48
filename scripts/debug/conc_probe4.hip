// Packed-fp32 VALU hazards beside another wave's MFMAs: explicit instruction sequences (one asm block each, so the order is
// exactly as written), each checked in-kernel against the value the sequence must produce.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/debug/conc_probe4.hip -o scripts/micro/bin/conc_probe4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void aggr_mfma(int iters, float* __restrict__ out) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.001f * (threadIdx.x + e)); y[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[i], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 12345.678f) out[0] = r;
}
// counts[t] = lanes x iterations in which test t produced a wrong value; lanes_hi[t] = of those, lanes >= 32
__global__ __launch_bounds__(256) void victim(int iters, unsigned* __restrict__ counts) {
    const float fa = 1.f + (float)(threadIdx.x & 63), fb = 100.f + (float)blockIdx.x;
    unsigned bad[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool upper = (threadIdx.x & 63) >= 32;
    for (int it = 0; it < iters; ++it) {
        const f32x2 a = {fa + (float)it, fa * 2.f}, b = {fb, fb + 0.5f};
        const float c = 7.f + (float)it, d = 0.25f;
        float r; f32x2 r2;
        // 0. WAW: packed write of v[40:41], then a plain write of v41 right behind it; v41 must hold c + d
        asm volatile("v_pk_add_f32 v[40:41], %1, %2\n\tv_add_f32 v41, %3, %4\n\ts_nop 7\n\tv_mov_b32 %0, v41" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41");
        if (r != c + d) { ++bad[0]; hi[0] += upper; }
        // 1. WAW with one independent instruction between
        asm volatile("v_pk_add_f32 v[40:41], %1, %2\n\tv_mov_b32 v42, %3\n\tv_add_f32 v41, %3, %4\n\ts_nop 7\n\tv_mov_b32 %0, v41" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41", "v42");
        if (r != c + d) { ++bad[1]; hi[1] += upper; }
        // 2. WAR: packed read of v[40:41], then plain writes of v40 / v41 right behind it
        asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\ts_nop 7\n\tv_pk_add_f32 %0, v[40:41], %3\n\tv_mov_b32 v40, %4\n\tv_mov_b32 v41, %4\n\ts_nop 7"
                     : "=&v"(r2) : "v"(a[0]), "v"(a[1]), "v"(b), "v"(c) : "v40", "v41");
        if (r2[0] != a[0] + b[0] || r2[1] != a[1] + b[1]) { ++bad[2]; hi[2] += upper; }
        // 3. RAW, packed -> plain with the compiler's one wait state (s_nop 0)
        asm volatile("v_pk_add_f32 v[40:41], %1, %2\n\ts_nop 0\n\tv_mul_f32 %0, %3, v41" : "=v"(r) : "v"(a), "v"(b), "v"(d) : "v40", "v41");
        if (r != d * (a[1] + b[1])) { ++bad[3]; hi[3] += upper; }
        // 4. RAW, packed -> plain with no wait state
        asm volatile("v_pk_add_f32 v[40:41], %1, %2\n\tv_mul_f32 %0, %3, v41" : "=v"(r) : "v"(a), "v"(b), "v"(d) : "v40", "v41");
        if (r != d * (a[1] + b[1])) { ++bad[4]; hi[4] += upper; }
        // 5. RAW, packed -> packed, no wait state
        asm volatile("v_pk_add_f32 v[40:41], %1, %2\n\tv_pk_add_f32 %0, v[40:41], %2" : "=v"(r2) : "v"(a), "v"(b) : "v40", "v41");
        if (r2[0] != (a[0] + b[0]) + b[0] || r2[1] != (a[1] + b[1]) + b[1]) { ++bad[5]; hi[5] += upper; }
        // 6. RAW, plain -> packed, no wait state
        asm volatile("v_add_f32 v40, %1, %2\n\tv_add_f32 v41, %1, %3\n\tv_pk_add_f32 %0, v[40:41], %4" : "=v"(r2) : "v"(c), "v"(d), "v"(fa), "v"(b) : "v40", "v41");
        if (r2[0] != (c + d) + b[0] || r2[1] != (c + fa) + b[1]) { ++bad[6]; hi[6] += upper; }
        // 7. WAW the other way: plain write of v41, packed write of v[40:41] right behind it
        asm volatile("v_add_f32 v41, %3, %4\n\tv_pk_add_f32 v[40:41], %1, %2\n\ts_nop 7\n\tv_mov_b32 %0, v41" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(d) : "v40", "v41");
        if (r != a[1] + b[1]) { ++bad[7]; hi[7] += upper; }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) if (bad[t]) { atomicAdd(&counts[t], bad[t]); atomicAdd(&counts[8 + t], hi[t]); }
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned* counts; float* sink; CK(hipMalloc(&counts, 64)); CK(hipMalloc(&sink, 64));
    const char* names[8] = {"WAW  pk write, plain write behind it", "WAW  pk write, 1 instr, plain write", "WAR  pk read, plain writes behind it", "RAW  pk -> plain, s_nop 0",
                            "RAW  pk -> plain, no wait", "RAW  pk -> pk, no wait", "RAW  plain -> pk, no wait", "WAW  plain write, pk write behind it"};
    for (int trial = 0; trial < 3; ++trial) {
        CK(hipMemset(counts, 0, 64)); CK(hipDeviceSynchronize());
        if (trial) for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(aggr_mfma, dim3(1024), dim3(256), 0, sa, 2000, sink);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim, dim3(4000), dim3(256), 0, sb, 200, counts);
        CK(hipDeviceSynchronize());
        unsigned h[16]; CK(hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost));
        printf("== trial %d (%s): wrong lane-results of %lld per test (of which lanes 32-63)\n", trial, trial ? "beside the MFMA kernel" : "alone", 4ll * 4000 * 256 * 200);
        for (int t = 0; t < 8; ++t) printf("   %d %-40s: %u (%u)\n", t, names[t], h[t], h[8 + t]);
    }
    return 0;
}
