// Which resource of a victim wave is corrupted when it shares a CU with another kernel, and which property of the
// aggressor does it?  Standalone (links libavexhip.so for the real fbank / GEMM kernels; everything else is synthetic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scripts/debug/conc_probe.hip -o scripts/micro/bin/conc_probe \
//         -L avex_amd/lib -lavexhip -Wl,-rpath,'$ORIGIN/../../../avex_amd/lib'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "avexhip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ victims
// NV live VGPR values per lane held across a spin; report = {mismatches, first bad index, value seen, block}
template <int NV>
__global__ __launch_bounds__(256) void vgpr_canary(int iters, unsigned* __restrict__ report) {
    unsigned v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = 0xA5000000u ^ (unsigned)(i << 16) ^ (blockIdx.x << 8 & 0xff00u) ^ threadIdx.x;
        asm volatile("" : "+v"(v[i]));
    }
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_sleep 4");
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned want = 0xA5000000u ^ (unsigned)(i << 16) ^ (blockIdx.x << 8 & 0xff00u) ^ threadIdx.x;
        if (v[i] != want) {
            if (atomicAdd(&report[0], 1u) == 0) { report[1] = (unsigned)i; report[2] = v[i]; report[3] = blockIdx.x; report[4] = threadIdx.x; }
        }
    }
}

// LDS pattern (static, BYTES) re-checked for a while
template <int WORDS>
__global__ __launch_bounds__(256) void lds_canary(int iters, unsigned* __restrict__ report) {
    __shared__ __attribute__((aligned(16))) unsigned words[WORDS];
    auto want = [&](int i) { return 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 16); };
    for (int i = threadIdx.x; i < WORDS; i += 256) words[i] = want(i);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < WORDS; i += 256) {
            const unsigned x = words[i];
            if (x != want(i)) {
                if (atomicAdd(&report[0], 1u) == 0) { report[1] = (unsigned)i; report[2] = x; report[3] = blockIdx.x; report[4] = threadIdx.x; }
                words[i] = want(i);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ synthetic aggressors
// (a) AGPR traffic only: 64 accumulation registers written and read back in a loop
__global__ __launch_bounds__(256) void aggr_agpr(int iters, float* __restrict__ out) {
    float a[64];
    float s = (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a[i]) : "v"(s + (float)i));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            float t;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a[i]));
            t = t * 1.0001f + 1.0f;
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a[i]) : "v"(t));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a[i])); r += t; }
    if (r == 12345.678f) out[0] = r;
}

// (b) MFMA with the accumulators wherever the compiler puts them for a 256-thread block (AGPRs) ...
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

// (c) 64 KiB of DYNAMIC LDS written and read with 16-byte accesses, no matrix instructions
__global__ __launch_bounds__(256) void aggr_dynlds(int iters, int bytes, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* p = (uint4*)smem;
    const int n = bytes / 16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < n; i += 256) p[i] = make_uint4(it, i, threadIdx.x, blockIdx.x);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) { const uint4 v = p[(i + 64) % n]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
        __syncthreads();
    }
    if (acc.x == 0x12345u && acc.y == 77u) out[0] = 1.f;
}

// ------------------------------------------------------------------ harness
struct Ctx {
    hipStream_t sa, sb;
    unsigned* report;
    float* sink;
    // real kernels
    avexhip_fbank_plan* plan;
    float* wav; float* fb_ref; float* fb_out; int B; int64_t T; int frames;
    void *A, *W, *O; int M;
};

static void run_aggr(Ctx& c, int which, int reps) {
    for (int r = 0; r < reps; ++r) {
        switch (which) {
            case 0: break;
            case 1: case 3: {
                avexhip_gemm_args g; memset(&g, 0, sizeof(g));
                g.A = c.A; g.lda = 768; g.W = c.W; g.ldw = 768; g.M = c.M; g.N = 2304; g.K = 768; g.out_half = c.O; g.ldh = 2304; g.variant = which;
                if (avexhip_gemm(&g, AVEXHIP_F16, c.sa) != 0) { printf("gemm failed: %s\n", avexhip_last_error()); exit(1); }
                break; }
            case 10: hipLaunchKernelGGL(aggr_agpr, dim3(1024), dim3(256), 0, c.sa, 400, c.sink); break;
            case 11: hipLaunchKernelGGL(aggr_mfma<256>, dim3(1024), dim3(256), 0, c.sa, 2000, c.sink); break;
            case 12: hipLaunchKernelGGL(aggr_mfma<1024>, dim3(1024), dim3(256), 0, c.sa, 2000, c.sink); break;
            case 13: hipLaunchKernelGGL(aggr_dynlds, dim3(1024), dim3(256), 65536, c.sa, 20, 65536, c.sink); break;
        }
    }
}
static const char* aggr_name(int w) {
    switch (w) {
        case 0: return "nothing"; case 1: return "gemm v1 (128-tile, reg staging, AGPR acc)"; case 3: return "gemm v3 (128-tile, LDS-DMA, AGPR acc)";
        case 10: return "synthetic: AGPR read/write only"; case 11: return "synthetic: MFMA, launch_bounds 256";
        case 12: return "synthetic: MFMA, launch_bounds 1024 (VGPR acc)"; case 13: return "synthetic: 64 KiB dynamic LDS traffic";
    }
    return "?";
}

int main() {
    Ctx c;
    CK(hipStreamCreateWithFlags(&c.sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c.sb, hipStreamNonBlocking));
    CK(hipMalloc(&c.report, 64)); CK(hipMalloc(&c.sink, 64));
    // fbank plan with a plain Hann window and a crude triangular bank (bit-compare against its own serial result)
    {
        std::vector<float> win(400), mel(257 * 128, 0.f);
        for (int i = 0; i < 400; ++i) win[i] = 0.5f - 0.5f * cosf(2.f * (float)M_PI * i / 399.f);
        for (int m = 0; m < 128; ++m) for (int k = 2 * m + 1; k <= 2 * m + 3 && k < 257; ++k) mel[k * 128 + m] = 1.f - fabsf((float)(k - 2 * m - 2)) * 0.5f;
        avexhip_fbank_config fc = {400, 160, 128, 32768.f, 0.97f, 1, 1.1920929e-07f, 15.41663f, 13.11164f};
        c.plan = avexhip_fbank_plan_create(&fc, win.data(), mel.data());
        if (!c.plan) { printf("plan: %s\n", avexhip_last_error()); return 1; }
        c.B = 32; c.T = 160000; c.frames = avexhip_fbank_num_frames(c.plan, c.T);
        std::vector<float> h((size_t)c.B * c.T);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f - 1.f) * 0.1f; }
        CK(hipMalloc(&c.wav, h.size() * 4)); CK(hipMemcpy(c.wav, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const size_t on = (size_t)c.B * c.frames * 128;
        CK(hipMalloc(&c.fb_ref, on * 4)); CK(hipMalloc(&c.fb_out, on * 4));
        if (avexhip_fbank_forward(c.plan, c.wav, c.B, c.T, c.T, c.fb_ref, nullptr) != 0) { printf("fbank: %s\n", avexhip_last_error()); return 1; }
        CK(hipDeviceSynchronize());
    }
    c.M = 32 * 496;
    {
        std::vector<_Float16> a((size_t)c.M * 768), w((size_t)2304 * 768);
        unsigned s = 777u;
        for (auto& v : a) { s = s * 1664525u + 1013904223u; v = (_Float16)(((float)(s >> 8) / 8388608.f - 1.f)); }
        for (auto& v : w) { s = s * 1664525u + 1013904223u; v = (_Float16)(((float)(s >> 8) / 8388608.f - 1.f) * 0.05f); }
        CK(hipMalloc(&c.A, a.size() * 2)); CK(hipMalloc(&c.W, w.size() * 2)); CK(hipMalloc(&c.O, (size_t)c.M * 2304 * 2));
        CK(hipMemcpy(c.A, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(c.W, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    const int aggrs[] = {0, 1, 3, 10, 11, 12, 13};
    const size_t on = (size_t)c.B * c.frames * 128;
    std::vector<float> href(on), hout(on);
    CK(hipMemcpy(href.data(), c.fb_ref, on * 4, hipMemcpyDeviceToHost));
    for (int ag : aggrs) {
        printf("== aggressor: %s\n", aggr_name(ag));
        // victim 1: real fbank
        {
            long wrong = 0; int launches = 0;
            for (int round = 0; round < 12; ++round) {
                CK(hipDeviceSynchronize());
                run_aggr(c, ag, ag >= 10 ? 6 : 30);
                for (int k = 0; k < 5; ++k) {
                    if (avexhip_fbank_forward(c.plan, c.wav, c.B, c.T, c.T, c.fb_out, c.sb) != 0) { printf("fbank: %s\n", avexhip_last_error()); return 1; }
                    CK(hipStreamSynchronize(c.sb));
                    CK(hipMemcpy(hout.data(), c.fb_out, on * 4, hipMemcpyDeviceToHost));
                    ++launches;
                    for (size_t i = 0; i < on; ++i) wrong += memcmp(&hout[i], &href[i], 4) != 0;
                }
            }
            CK(hipDeviceSynchronize());
            printf("   fbank           : %ld wrong elements over %d overlapped launches\n", wrong, launches);
        }
        // victims 2..: canaries
        auto canary = [&](const char* name, auto launch) {
            unsigned tot = 0, first[5] = {0, 0, 0, 0, 0};
            for (int round = 0; round < 6; ++round) {
                CK(hipMemset(c.report, 0, 64));
                CK(hipDeviceSynchronize());
                run_aggr(c, ag, ag >= 10 ? 6 : 30);
                for (int k = 0; k < 3; ++k) launch();
                CK(hipDeviceSynchronize());
                unsigned r[5];
                CK(hipMemcpy(r, c.report, 20, hipMemcpyDeviceToHost));
                if (r[0] && !tot) memcpy(first, r, 20);
                tot += r[0];
            }
            printf("   %-16s: %u mismatches", name, tot);
            if (tot) printf("  (first: index %u value %#x [as float %g] block %u thread %u)", first[1], first[2], *(float*)&first[2], first[3], first[4]);
            printf("\n");
        };
        canary("vgpr_canary<100>", [&] { hipLaunchKernelGGL(vgpr_canary<100>, dim3(4000), dim3(256), 0, c.sb, 300, c.report); });
        canary("vgpr_canary<52>", [&] { hipLaunchKernelGGL(vgpr_canary<52>, dim3(4000), dim3(256), 0, c.sb, 300, c.report); });
        canary("vgpr_canary<40>", [&] { hipLaunchKernelGGL(vgpr_canary<40>, dim3(4000), dim3(256), 0, c.sb, 300, c.report); });
        canary("lds_canary 31KB", [&] { hipLaunchKernelGGL(lds_canary<7744>, dim3(4000), dim3(256), 0, c.sb, 60, c.report); });
    }
    printf("done\n");
    return 0;
}
