// Which instruction class of the fbank kernel goes wrong beside a pure-MFMA wave?  Each victim computes a fixed function of
// (block, thread) and stores it; the result beside the aggressor is bit-compared with the result of a serial launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/debug/conc_probe2.hip -o scripts/micro/bin/conc_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int LB>
__global__ __launch_bounds__(LB) void aggr_mfma(int iters, float* __restrict__ out) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 x, y;
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

__device__ __forceinline__ float seed(int i) { return 0.001f * (float)((threadIdx.x * 7 + blockIdx.x * 13 + i * 31) % 1000) - 0.5f; }

// packed fp32 chain (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
__global__ __launch_bounds__(256) void vic_pk(int iters, float* __restrict__ out) {
    f32x2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){seed(i), seed(i + 8)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2 t;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"((f32x2){0.5f, 0.4999f}));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i][0] - a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// the same arithmetic with scalar fp32 instructions
__global__ __launch_bounds__(256) void vic_f32(int iters, float* __restrict__ out) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed(i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(a[(i + 1) & 15]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(t), "v"(0.4999f));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// 64-bit moves
__global__ __launch_bounds__(256) void vic_mov64(int iters, float* __restrict__ out) {
    f32x2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){seed(i), seed(i + 8)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2 t;
            asm volatile("v_mov_b64 %0, %1" : "=v"(t) : "v"(a[(i + 3) & 7]));
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(a[i]) : "v"(t), "v"(a[i]));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i][0] * 3.f - a[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// LDS exchange through a wave-private buffer with only s_waitcnt lgkmcnt(0) between phases (the fbank passes' pattern)
__global__ __launch_bounds__(256) void vic_lds(int iters, float* __restrict__ out) {
    __shared__ float2 z[4][8 * 72];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = make_float2(seed(i), seed(i + 8));
    float2* zz = z[wave];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) zz[k * 72 + lane] = x[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int k1 = lane >> 3, m2 = lane & 7;
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) x[m1] = zz[k1 * 72 + 8 * m1 + m2];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i].x * (float)(i + 1) - x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
// ds_bpermute butterfly sums
__global__ __launch_bounds__(256) void vic_bperm(int iters, float* __restrict__ out) {
    float s = seed(0), tot = 0.f;
    for (int it = 0; it < iters; ++it) {
        float v = s + (float)it;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        tot += v * 1e-3f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = tot;
}
// transcendental / division
__global__ __launch_bounds__(256) void vic_trans(int iters, float* __restrict__ out) {
    float a = fabsf(seed(0)) + 1.f, tot = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float q = sqrtf(a + (float)it);
        const float l = logf(q + 1.f);
        tot += l / (q + 0.5f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = tot;
}
// predicated global loads (exec-masked branches) + twiddle-like gather from an LDS table
__global__ __launch_bounds__(256) void vic_gather(int iters, const float2* __restrict__ tab, float* __restrict__ out) {
    __shared__ float2 tw[512];
    tw[threadIdx.x] = tab[threadIdx.x]; tw[threadIdx.x + 256] = tab[threadIdx.x + 256];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float2 acc = make_float2(0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float2 t = tw[(lane * k + it) & 511]; acc.x += t.x * (float)k; acc.y -= t.y; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int NB = 4000;
    float *out, *ref, *sink; float2* tab;
    CK(hipMalloc(&out, NB * 256 * 4)); CK(hipMalloc(&ref, NB * 256 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&tab, 512 * 8));
    std::vector<float2> ht(512);
    for (int k = 0; k < 512; ++k) ht[k] = make_float2(cosf(0.0122718f * k), -sinf(0.0122718f * k));
    CK(hipMemcpy(tab, ht.data(), 512 * 8, hipMemcpyHostToDevice));
    std::vector<float> ho(NB * 256), hr(NB * 256);
    struct V { const char* name; int id; };
    const V vs[] = {{"packed fp32 (v_pk_add/mul_f32)", 0}, {"scalar fp32 (v_add/mul_f32)", 1}, {"v_mov_b64 / v_pk_mov_b32", 2}, {"LDS exchange, wave-local waits", 3},
                    {"ds_bpermute sums", 4}, {"sqrt / log / divide", 5}, {"LDS table gather", 6}};
    auto launch = [&](int id, float* o, hipStream_t s) {
        switch (id) {
            case 0: hipLaunchKernelGGL(vic_pk, dim3(NB), dim3(256), 0, s, 400, o); break;
            case 1: hipLaunchKernelGGL(vic_f32, dim3(NB), dim3(256), 0, s, 400, o); break;
            case 2: hipLaunchKernelGGL(vic_mov64, dim3(NB), dim3(256), 0, s, 400, o); break;
            case 3: hipLaunchKernelGGL(vic_lds, dim3(NB), dim3(256), 0, s, 100, o); break;
            case 4: hipLaunchKernelGGL(vic_bperm, dim3(NB), dim3(256), 0, s, 200, o); break;
            case 5: hipLaunchKernelGGL(vic_trans, dim3(NB), dim3(256), 0, s, 200, o); break;
            case 6: hipLaunchKernelGGL(vic_gather, dim3(NB), dim3(256), 0, s, 100, tab, o); break;
        }
    };
    for (int lb = 0; lb < 2; ++lb) {
        printf("== aggressor: pure MFMA loop, launch_bounds %d\n", lb ? 1024 : 256);
        for (const V& v : vs) {
            launch(v.id, ref, sb);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hr.data(), ref, NB * 256 * 4, hipMemcpyDeviceToHost));
            long wrong = 0; int first = -1;
            for (int round = 0; round < 8; ++round) {
                CK(hipDeviceSynchronize());
                for (int r = 0; r < 6; ++r) {
                    if (lb) hipLaunchKernelGGL(aggr_mfma<1024>, dim3(1024), dim3(256), 0, sa, 2000, sink);
                    else hipLaunchKernelGGL(aggr_mfma<256>, dim3(1024), dim3(256), 0, sa, 2000, sink);
                }
                for (int k = 0; k < 3; ++k) {
                    CK(hipMemsetAsync(out, 0xff, NB * 256 * 4, sb));
                    launch(v.id, out, sb);
                    CK(hipStreamSynchronize(sb));
                    CK(hipMemcpy(ho.data(), out, NB * 256 * 4, hipMemcpyDeviceToHost));
                    for (int i = 0; i < NB * 256; ++i) if (memcmp(&ho[i], &hr[i], 4)) { if (first < 0) first = i; ++wrong; }
                }
            }
            printf("   %-34s: %ld wrong of %d", v.name, wrong, 24 * NB * 256);
            if (first >= 0) printf("  (first at block %d thread %d: got %g want %g)", first / 256, first % 256, ho[first], hr[first]);
            printf("\n");
        }
    }
    printf("done\n");
    return 0;
}
