// Packed-fp32 instructions with op_sel (half-swapping) operands: write-after-read by the NEXT packed instruction, beside
// another wave's MFMAs.  Sequences lifted from the compiled radix-8 butterfly of fbank_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/debug/conc_probe5.hip -o scripts/micro/bin/conc_probe5
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
#define NT 10
// PRE loads v[40:41] = a, v[44:45] = b, v[46:47] = c, v[48:49] = d and lets everything settle
#define PRE "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v44, %4\n\tv_mov_b32 v45, %5\n\tv_mov_b32 v46, %6\n\tv_mov_b32 v47, %7\n\tv_mov_b32 v48, %8\n\tv_mov_b32 v49, %9\n\ts_nop 7\n\t"
#define POST "\n\ts_nop 7\n\tv_mov_b32 %0, v42\n\tv_mov_b32 %1, v43"
#define IO : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(d0), "v"(d1) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49"
__global__ __launch_bounds__(256) void victim(int iters, unsigned* __restrict__ counts, float* __restrict__ seen) {
    const float l = (float)(threadIdx.x & 63);
    unsigned bad[NT], hi[NT];
    for (int t = 0; t < NT; ++t) { bad[t] = 0; hi[t] = 0; }
    const bool upper = (threadIdx.x & 63) >= 32;
    for (int it = 0; it < iters; ++it) {
        const float a0 = 1000.f + l, a1 = 2000.f + l, b0 = 3.f + (float)it, b1 = 5.f, c0 = 100000.f, c1 = 200000.f, d0 = 300000.f, d1 = 400000.f;
        float r0, r1;
#define CHECK(t, e0, e1) if (r0 != (e0) || r1 != (e1)) { if (!bad[t] && atomicAdd(&counts[2 * NT + t], 1u) == 0) { seen[4 * t] = r0; seen[4 * t + 1] = r1; seen[4 * t + 2] = (e0); seen[4 * t + 3] = (e1); } ++bad[t]; hi[t] += upper; }
        // 0: swapping add reads v[40:41]; the NEXT swapping add overwrites v[40:41]   (the compiled sequence)
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 v[40:41], v[46:47], v[48:49] op_sel:[0,1] op_sel_hi:[1,0]" POST IO);
        CHECK(0, a0 + b1, a1 + b0)
        // 1: the same without half swapping on either
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45]\n\tv_pk_add_f32 v[40:41], v[46:47], v[48:49]" POST IO);
        CHECK(1, a0 + b0, a1 + b1)
        // 2: reader swaps, writer does not
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 v[40:41], v[46:47], v[48:49]" POST IO);
        CHECK(2, a0 + b1, a1 + b0)
        // 3: reader plain, writer swaps
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45]\n\tv_pk_add_f32 v[40:41], v[46:47], v[48:49] op_sel:[0,1] op_sel_hi:[1,0]" POST IO);
        CHECK(3, a0 + b0, a1 + b1)
        // 4: reader swaps, plain 32-bit writers behind it
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v40, v46\n\tv_mov_b32 v41, v47" POST IO);
        CHECK(4, a0 + b1, a1 + b0)
        // 5: the swapped SOURCE (src1 = v[44:45]) overwritten by the next packed instruction
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 v[44:45], v[46:47], v[48:49]" POST IO);
        CHECK(5, a0 + b1, a1 + b0)
        // 6: one independent instruction between reader and writer
        asm volatile(PRE "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v50, v46\n\tv_pk_add_f32 v[40:41], v[46:47], v[48:49]" POST IO, "v50");
        CHECK(6, a0 + b1, a1 + b0)
        // 7: the whole compiled group: two swapping subtracts, then two swapping adds, the last one overwriting the source of the one before
        asm volatile(PRE "v_pk_add_f32 v[50:51], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                         "v_pk_add_f32 v[52:53], v[46:47], v[48:49] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                         "v_pk_add_f32 v[42:43], v[40:41], v[44:45] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 v[40:41], v[46:47], v[48:49] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 v[54:55], v[50:51], v[52:53] neg_lo:[0,1] neg_hi:[0,1]" POST IO, "v50", "v51", "v52", "v53", "v54", "v55");
        CHECK(7, a0 + b1, a1 + b0)
        // 8: RAW through a swapping reader with no wait: plain packed write, swapping read right behind
        asm volatile(PRE "v_pk_add_f32 v[50:51], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[50:51], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]" POST IO, "v50", "v51");
        CHECK(8, (a0 + b0) + c1, (a1 + b1) + c0)
        // 9: RAW, swapping read of the SWAPPED operand right behind its packed write
        asm volatile(PRE "v_pk_add_f32 v[50:51], v[40:41], v[44:45]\n\tv_pk_add_f32 v[42:43], v[46:47], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]" POST IO, "v50", "v51");
        CHECK(9, c0 + (a1 + b1), c1 + (a0 + b0))
    }
    for (int t = 0; t < NT; ++t) if (bad[t]) { atomicAdd(&counts[t], bad[t]); atomicAdd(&counts[NT + t], hi[t]); }
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned* counts; float *sink, *seen; CK(hipMalloc(&counts, 4 * 3 * NT)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&seen, 16 * NT));
    const char* names[NT] = {"WAR swap-read, swap-write next", "WAR plain-read, plain-write next", "WAR swap-read, plain pk write next", "WAR plain-read, swap-write next",
                             "WAR swap-read, v_mov writes next", "WAR on the swapped source", "WAR swap-read, 1 instr, pk write", "compiled group of five", "RAW pk write -> swap-read src0",
                             "RAW pk write -> swap-read src1"};
    for (int trial = 0; trial < 3; ++trial) {
        CK(hipMemset(counts, 0, 4 * 3 * NT)); CK(hipDeviceSynchronize());
        if (trial) for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(aggr_mfma, dim3(1024), dim3(256), 0, sa, 2000, sink);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim, dim3(4000), dim3(256), 0, sb, 100, counts, seen);
        CK(hipDeviceSynchronize());
        unsigned h[3 * NT]; float hs[4 * NT]; CK(hipMemcpy(h, counts, 4 * 3 * NT, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, seen, 16 * NT, hipMemcpyDeviceToHost));
        printf("== trial %d (%s): wrong lane-results of %lld per test (of which lanes 32-63)\n", trial, trial ? "beside the MFMA kernel" : "alone", 4ll * 4000 * 256 * 100);
        for (int t = 0; t < NT; ++t) {
            printf("   %d %-36s: %u (%u)", t, names[t], h[t], h[NT + t]);
            if (h[t]) printf("   e.g. got (%g, %g) want (%g, %g)", hs[4 * t], hs[4 * t + 1], hs[4 * t + 2], hs[4 * t + 3]);
            printf("\n");
        }
    }
    return 0;
}
