// Does the MFMA's power (and so, under the 1400 W cap, its sustained rate) depend on how many mantissa bits of an operand are in use?
// Register-only loop of v_mfma_f32_16x16x32_f16 as in mfma_power.hip; the B operand (the activations in gemm256p_kernel) is random
// halves whose low `drop` mantissa bits are zero (drop = 3: the 8-bit significand of bf16, kept in f16 format), the A operand (weights)
// keeps all 11 bits.  Also: both masked, bf16 MFMA on random bf16, and Gaussian-distributed values instead of uniform ones.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_mantissa.hip -o /tmp/mfma_mantissa && /tmp/mfma_mantissa
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned short u8v __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline float urand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return ((int)((s >> 9) & 0xffff) - 32768) * (1.0f / 32768.0f);
}
__device__ inline float grand(unsigned& s) {      // sum of four uniforms: bell-shaped, unit-ish variance
    return (urand(s) + urand(s) + urand(s) + urand(s)) * 0.87f;
}
__device__ inline h8 rnd8(unsigned s, int gauss, int drop) {
    h8 v;
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(gauss ? grand(s) : urand(s));
    if (drop > 0) {
        u8v bits = __builtin_bit_cast(u8v, v);
        const unsigned short half = (unsigned short)(1u << (drop - 1)), mask = (unsigned short)(0xffffu << drop);
        for (int i = 0; i < 8; ++i) bits[i] = (unsigned short)((bits[i] + half) & mask);     // round to nearest (ties up), in place
        v = __builtin_bit_cast(h8, bits);
    }
    return v;
}

template <int BF>
__global__ __launch_bounds__(512) void loop(int iters, float* out, int gauss, int drop_a, int drop_b) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = rnd8(threadIdx.x * 977u + i * 131u + blockIdx.x, gauss, drop_a);
        b[i] = rnd8(threadIdx.x * 613u + i * 257u + 7u, gauss, drop_b);
    }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f4){0.f, 0.f, 0.f, 0.f};
    if (BF) {
        b8 ab[4], bb[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 8; ++e) { ab[i][e] = (__bf16)(float)a[i][e]; bb[i][e] = (__bf16)(float)b[i][e]; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i % 4], bb[(i + 1) % 4], c[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i % 4], b[(i + 1) % 4], c[i], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (sum == 123.456f) out[0] = sum;
}

template <int BF>
static void run(const char* name, int gauss, int drop_a, int drop_b) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 1 << 19;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0, last = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        loop<BF><<<256, 512>>>(iters, out, gauss, drop_a, drop_b);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        last = 256.0 * 8 * iters * 8 * 16384.0 / ms / 1e9;
        if (rep > 0 && last > best) best = last;
    }
    std::printf("%-58s last %7.1f  best(rep>0) %7.1f TFLOP/s\n", name, last, best);
    hipFree(out);
}

int main() {
    for (int round = 0; round < 2; ++round) {
        run<0>("f16 uniform, 11-bit A, 11-bit B", 0, 0, 0);
        run<0>("f16 uniform, 11-bit A,  8-bit B (drop 3)", 0, 0, 3);
        run<0>("f16 uniform, 11-bit A,  6-bit B (drop 5)", 0, 0, 5);
        run<0>("f16 uniform, 11-bit A,  4-bit B (drop 7)", 0, 0, 7);
        run<0>("f16 uniform,  8-bit A,  8-bit B", 0, 3, 3);
        run<1>("bf16 uniform", 0, 0, 0);
        run<0>("f16 gaussian, 11-bit A, 11-bit B", 1, 0, 0);
        run<0>("f16 gaussian, 11-bit A,  8-bit B (drop 3)", 1, 0, 3);
        run<1>("bf16 gaussian", 1, 0, 0);
    }
    return 0;
}
