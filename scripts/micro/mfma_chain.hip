// Cost of back-to-back v_mfma_f32_32x32x16_f16 on gfx950 when consecutive instructions share the accumulator (a dependent chain
// through SrcC) against round-robin over 2 / 4 accumulators; one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_chain.hip -o scripts/micro/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int NACC>
__global__ __launch_bounds__(512) void chain_kernel(int iters, unsigned long long* __restrict__ cyc, float* __restrict__ sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][5];
    if (r == 1234.5f) sink[0] = r;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    unsigned long long* cyc; float* sink;
    CK(hipMalloc(&cyc, 256)); CK(hipMalloc(&sink, 64));
    const int iters = 2000;
    for (int nw = 1; nw <= 2; ++nw) {
        unsigned long long h[8];
#define RUN(N) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(chain_kernel<N>, dim3(256), dim3(256 * nw), 0, 0, iters, cyc, sink); CK(hipDeviceSynchronize()); } \
        CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); printf("%d wave(s) per SIMD, %d accumulator(s) round robin: %.1f cycles per MFMA (wave 0), %.1f (last wave)\n", nw, N, (double)h[0] / iters / 16, (double)h[4 * nw - 1] / iters / 16);
        RUN(1) RUN(2) RUN(4)
    }
    return 0;
}
