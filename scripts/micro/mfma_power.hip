// Register-only MFMA loop under the board power cap: which f16 MFMA shape sustains more FLOP/s when power, not issue rate, is the
// limit?  No LDS, no memory traffic; operands are pseudo-random halves (toggle rate matters for power).  Build & run:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline h8 rnd8(unsigned s) {
    h8 v;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = (_Float16)(((int)((s >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
    return v;
}

// MODE 0: v_mfma_f32_16x16x32_f16, 8 independent accumulators (32 regs); MODE 1: v_mfma_f32_32x32x16_f16, 4 accumulators (64 regs)
// PAT 0: A and B both change between consecutive MFMAs; 1: A fixed, B changes; 2: neither changes (accumulators still differ)
template <int MODE, int NOPS, int PAT>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* out, int zero_data) {
    h8 a[NOPS], b[NOPS];
    for (int i = 0; i < NOPS; ++i) {
        a[i] = rnd8(threadIdx.x * 977u + i * 131u + blockIdx.x);
        b[i] = rnd8(threadIdx.x * 613u + i * 257u + 7u);
        if (zero_data) { a[i] = (h8)(_Float16)0.f; b[i] = (h8)(_Float16)0.f; }
    }
    float sum = 0.f;
    if (MODE == 0) {
        f4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[PAT == 0 ? i % NOPS : 0], b[PAT == 2 ? 0 : (i + 1) % NOPS], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) sum += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PAT == 0 ? i % NOPS : 0], b[PAT == 2 ? 0 : (i + 1) % NOPS], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) sum += c[i][r];
    }
    if (sum == 123.456f) out[0] = sum;
}

// 16x16x32 loop whose operands are re-read from LDS every iteration: NLD ds_read_b128 per 8 MFMAs per wave (gemm256_kernel reads 6 per 8:
// a 128 x 64 wave tile takes 8 A + 4 B fragments per 32 MFMAs; a 128 x 128 wave tile would take 8 + 8 per 64 = 4 per 16).
template <int NLD>
__global__ __launch_bounds__(512) void mfma_lds_loop(int iters, float* out) {
    __shared__ h8 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = rnd8(i * 977u + blockIdx.x);
    __syncthreads();
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(threadIdx.x * 977u + i * 131u); b[i] = rnd8(threadIdx.x * 613u + i * 257u + 7u); }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const h8 v = lds[(idx + i * 512) & 4095];
            if (i & 1) b[(i >> 1) & 3] = v; else a[(i >> 1) & 3] = v;
        }
        idx = (idx + 64) & 4095;
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i % 4], b[(i + 1) % 4], c[i], 0, 0, 0);
    }
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (sum == 123.456f) out[0] = sum;
}

template <int NLD>
static void run_lds(const char* name) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 1 << 19;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        mfma_lds_loop<NLD><<<256, 512>>>(iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 8 * iters * 8 * 16384.0;
        std::printf("%-34s random rep %d: %8.2f ms  %7.1f TFLOP/s\n", name, rep, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

template <int MODE, int PAT = 0>
static void run(const char* name, int zero) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 1 << 19;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        mfma_loop<MODE, 4, PAT><<<256, 512>>>(iters, out, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * 8 * iters * (MODE == 0 ? 8 * 16384.0 : 4 * 32768.0);
        std::printf("%-34s %s rep %d: %8.2f ms  %7.1f TFLOP/s\n", name, zero ? "zeros " : "random", rep, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<0, 0>("16x16x32 registers only", 0);
    run_lds<2>("16x16x32 + 2 ds_read_b128 / 8 MFMA");
    run_lds<4>("16x16x32 + 4 ds_read_b128 / 8 MFMA");
    run_lds<6>("16x16x32 + 6 ds_read_b128 / 8 MFMA");
    run_lds<8>("16x16x32 + 8 ds_read_b128 / 8 MFMA");
    run<0, 0>("16x16x32 registers only", 0);
    return 0;
}
