// Which operand-select forms of the packed-fp32 instructions read a wrong value beside another wave's MFMAs (gfx950)?
// One instruction per test on settled registers (s_nop 7 on both sides): no dependency hazard is involved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/debug/conc_probe6.hip -o scripts/micro/bin/conc_probe6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(1024) void aggr(int iters, float* __restrict__ out) {
    if (KIND == 0) {          // MFMA 16x16x32 f16
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
    } else {                  // plain VALU FMAs, no matrix instruction
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        }
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += a[i];
        if (r == 12345.678f) out[0] = r;
    }
}
#define NT 16
#define PRE "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v44, %4\n\tv_mov_b32 v45, %5\n\tv_mov_b32 v46, %6\n\tv_mov_b32 v47, %7\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\ts_nop 7\n\t"
#define POST "\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\tv_mov_b32 %1, v43"
#define IO : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
__global__ __launch_bounds__(256) void victim(int iters, unsigned* __restrict__ counts, float* __restrict__ seen) {
    const float l = (float)(threadIdx.x & 63);
    unsigned bad[NT], hi[NT];
    for (int t = 0; t < NT; ++t) { bad[t] = 0; hi[t] = 0; }
    const bool upper = (threadIdx.x & 63) >= 32;
    for (int it = 0; it < iters; ++it) {
        const float a0 = 1024.f + l, a1 = 2048.f + l, b0 = 3.f + (float)(it & 7), b1 = 16.f, c0 = 65536.f, c1 = 131072.f;
        float r0, r1;
#define CHECK(t, e0, e1) if (r0 != (e0) || r1 != (e1)) { if (!bad[t] && atomicAdd(&counts[2 * NT + t], 1u) == 0) { seen[4 * t] = r0; seen[4 * t + 1] = r1; seen[4 * t + 2] = (e0); seen[4 * t + 3] = (e1); } ++bad[t]; hi[t] += upper; }
#define T1(t, INSTR, e0, e1) asm volatile(PRE INSTR POST IO); CHECK(t, e0, e1)
        T1(0, "v_pk_add_f32 v[42:43], v[40:41], v[44:45]", a0 + b0, a1 + b1)
        T1(1, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1]", a0 + b1, a1 + b1)                       // lo lane reads src1.hi
        T1(2, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[1,0]", a1 + b0, a1 + b1)                       // lo lane reads src0.hi
        T1(3, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel_hi:[1,0]", a0 + b0, a1 + b0)                    // hi lane reads src1.lo
        T1(4, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel_hi:[0,1]", a0 + b0, a0 + b1)                    // hi lane reads src0.lo
        T1(5, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[1,1] op_sel_hi:[0,0]", a1 + b1, a0 + b0)       // both swapped
        T1(6, "v_pk_mul_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1]", a0 * b1, a1 * b1)
        T1(7, "v_pk_fma_f32 v[42:43], v[40:41], v[44:45], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,1,1]", __builtin_fmaf(a0, b0, c1), __builtin_fmaf(a1, b1, c1))
        T1(8, "v_pk_fma_f32 v[42:43], v[40:41], v[44:45], v[46:47] op_sel:[0,1,0] op_sel_hi:[1,1,1]", __builtin_fmaf(a0, b1, c0), __builtin_fmaf(a1, b1, c1))
        T1(9, "v_pk_fma_f32 v[42:43], v[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,1]", __builtin_fmaf(a0, b0, c0), __builtin_fmaf(a1, b0, c1))
        T1(10, "v_pk_mov_b32 v[42:43], v[40:41], v[44:45] op_sel:[1,0]", a1, b0)
        T1(11, "v_pk_mov_b32 v[42:43], v[40:41], v[44:45] op_sel:[0,1]", a0, b1)
        T1(12, "v_pk_mov_b32 v[42:43], v[40:41], v[44:45] op_sel:[1,1]", a1, b1)
        T1(13, "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]", a0 - b1, a1 - b1)
        T1(14, "v_pk_add_f32 v[42:43], v[44:45], v[40:41] op_sel:[0,1]", b0 + a1, b1 + a1)                      // the half-swapped value varies per lane
        T1(15, "v_mov_b64 v[42:43], v[44:45]", b0, b1)
    }
    for (int t = 0; t < NT; ++t) if (bad[t]) { atomicAdd(&counts[t], bad[t]); atomicAdd(&counts[NT + t], hi[t]); }
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned* counts; float *sink, *seen; CK(hipMalloc(&counts, 4 * 3 * NT)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&seen, 16 * NT));
    const char* names[NT] = {"pk_add, no op_sel", "pk_add op_sel:[0,1]", "pk_add op_sel:[1,0]", "pk_add op_sel_hi:[1,0]", "pk_add op_sel_hi:[0,1]", "pk_add op_sel:[1,1] op_sel_hi:[0,0]",
                             "pk_mul op_sel:[0,1]", "pk_fma op_sel:[0,0,1]", "pk_fma op_sel:[0,1,0]", "pk_fma op_sel_hi:[1,0,1]", "pk_mov op_sel:[1,0]", "pk_mov op_sel:[0,1]", "pk_mov op_sel:[1,1]",
                             "pk_add op_sel:[0,1] + neg", "pk_add op_sel:[0,1], src1 per-lane", "v_mov_b64"};
    const char* tn[3] = {"alone", "beside an MFMA kernel", "beside a VALU-only kernel"};
    for (int trial = 0; trial < 3; ++trial) {
        CK(hipMemset(counts, 0, 4 * 3 * NT)); CK(hipDeviceSynchronize());
        if (trial == 1) for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(aggr<0>, dim3(1024), dim3(256), 0, sa, 2000, sink);
        if (trial == 2) for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(aggr<1>, dim3(1024), dim3(256), 0, sa, 2000, sink);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim, dim3(4000), dim3(256), 0, sb, 64, counts, seen);
        CK(hipDeviceSynchronize());
        unsigned h[3 * NT]; float hs[4 * NT]; CK(hipMemcpy(h, counts, 4 * 3 * NT, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, seen, 16 * NT, hipMemcpyDeviceToHost));
        printf("== %s: wrong lane-results of %lld per test (of which lanes 32-63)\n", tn[trial], 4ll * 4000 * 256 * 64);
        for (int t = 0; t < NT; ++t) {
            printf("   %2d %-38s: %u (%u)", t, names[t], h[t], h[NT + t]);
            if (h[t]) printf("   e.g. got (%g, %g) want (%g, %g)", hs[4 * t], hs[4 * t + 1], hs[4 * t + 2], hs[4 * t + 3]);
            printf("\n");
        }
    }
    return 0;
}
