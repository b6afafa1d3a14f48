// The K = 768 GEMM with ONE wave per SIMD: 256 x 256 tile, 4 waves x (128 x 128), 256 accumulator registers per lane (round 5, after
// profiles/r05d: under the power cap the K loop is 75 % of fc1's joules, and profiles/r01h_mfma_power.txt prices a ds_read_b128 at 0.55 of a
// 16x16x32 MFMA -- the product's 128 x 64 wave tile reads 3 fragments per 8 MFMAs, a 128 x 128 one reads 2).  A microbenchmark of that K loop
// with real LDS-DMA, swizzled fragment reads, register double buffering of the fragments (one k-step ahead) and one barrier per k-step;
// the epilogue is bias + GELU + f16 + lane-swap transposes + stores, serial (nothing else runs on a SIMD).  CHECKED against the host.
// Pipeline: k-steps of 32 (one MFMA depth), FOUR 32 KB stages: the request for step g + 4 goes out behind the barrier of step g and is
// needed at the barrier of step g + 3 -- three steps (3 072 MFMA cycles) of distance.  (First form, two 64 KB stages, one K-tile of
// distance: 445 - 495 us for the bare loop -- with one wave per SIMD every cycle of L2 -> LDS latency beyond the distance is exposed.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/micro/gemm_w4.hip -o scripts/micro/bin/gemm_w4
//   scripts/micro/bin/gemm_w4 check ;  scripts/micro/bin/gemm_w4 [iters]
#include "../../avex_amd/csrc/common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef W4_BAR_AT
#define W4_BAR_AT 8      // a k-step's barrier sits behind this many of its 64 MFMAs
#endif
#ifndef W4_KO
#define W4_KO 0          // 1: no epilogue (accumulators kept alive): the bare K loop; timing-only knock-outs: 2 no DMA in the loop, 4 no barrier / counted wait, 8 no fragment reads in the loop
#endif

constexpr int BK = 32, NK = 24, K = NK * BK;
constexpr int STAGE = 32768;                   // W tile 256 rows x 64 B (16 KiB) + X tile 256 rows x 64 B (16 KiB)
constexpr int NSTG = 4;
constexpr int BIAS_OFF = NSTG * STAGE;
constexpr int LDS_BYTES = BIAS_OFF + 12288;

#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename F> static __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
static __device__ __forceinline__ void w4_dma16(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
static __device__ __forceinline__ float relu_nc(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
struct TileCoord { int m0, n0; };

__global__ __launch_bounds__(256) void w4_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ out,
                                                 const float* __restrict__ bias, int M, int N, unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef f16x8 v8;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = N / 256, tiles_m = M / 256, ntiles = tiles_m * tiles_n;
    float* ldsbias = (float*)(smem + BIAS_OFF);
    for (int i = tid; i < N; i += 256) ldsbias[i] = bias[i];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    int nt_mine = 0;
    for (int it = 0; (it * 8 + xcd) * per_xcd + slot < ntiles; ++it) ++nt_mine;
    if (nt_mine == 0) return;
    auto coords = [&](int t_idx) __attribute__((always_inline)) -> TileCoord {
        const int t = (t_idx * 8 + xcd) * per_xcd + slot;
        const int per_group = 8 * tiles_n;
        const int gid = t / per_group;
        const int first_m = gid * 8;
        const int gsz = (tiles_m - first_m) < 8 ? (tiles_m - first_m) : 8;
        const int r = t - gid * per_group;
        const int tn = r / gsz, tm = first_m + (r - tn * gsz);
        return {tm * 256, tn * 256};
    };
    asm volatile("" : "+v"(lane));
    const int lc = lane & 15, lg = lane >> 4;
    // LDS-DMA: a piece is 16 rows x 64 B; piece p = wid + 4 q (q = 0 .. 3) of each operand's 16; 16-byte slot c of row r sits at slot
    // c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 of a 16-row block: 64-byte rows put rows r and r + 4 on the same banks).  Rows 64 apart
    // share their swizzle: ONE lane offset, a scalar base per piece
    const int r0 = 16 * wid + (lane >> 2);
    const unsigned doff = (unsigned)(r0 * K + (((lane & 3) ^ ((r0 >> 2) & 3)) << 3)) * 2u;
    const int foff = lc * 64 + ((lg ^ ((lc >> 2) & 3)) << 4);
    const unsigned aw = (unsigned)(foff + (128 * wm) * 64);                   // W fragments (stage 0)
    const unsigned ax = (unsigned)(foff + 16384 + (128 * wn) * 64);           // X fragments
    const unsigned st_voff = (unsigned)(lc * N * 2 + 16 * lg);
    const float* bias_lane = ldsbias + 128 * wm + 4 * lg;

    auto dma_piece = [&](const char* wb, const char* xb, int ks, int stg, int q) __attribute__((always_inline)) {      // piece q of W and of X
        const unsigned base = (unsigned)(stg * STAGE) + (unsigned)((wid + 4 * q) * 1024);
        w4_dma16(wb + ks * (BK * 2) + (int64_t)q * (64 * K * 2), doff, base);
        w4_dma16(xb + ks * (BK * 2) + (int64_t)q * (64 * K * 2), doff, base + 16384);
    };
    f32x4 acc[8][8];
    v8 fw[2][8], fx[2][8];                         // [buffer][16-row block]
    auto read_frags = [&](int buf, int stg) __attribute__((always_inline)) {
        const char* pw = smem + stg * STAGE + aw;
        const char* px = smem + stg * STAGE + ax;
#pragma unroll
        for (int i = 0; i < 8; ++i) fw[buf][i] = *(const v8*)(pw + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) fx[buf][j] = *(const v8*)(px + j * 1024);
    };

    TileCoord cur = coords(0), nxt = cur;
    const char* wb_cur = (const char*)(W + (int64_t)cur.n0 * K);
    const char* xb_cur = (const char*)(A + (int64_t)cur.m0 * K);
    const char *wb_nxt = wb_cur, *xb_nxt = xb_cur;
    __syncthreads();
    // prologue: steps 0 .. 3 requested; step 0 waited for and read
#pragma unroll
    for (int g = 0; g < NSTG; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(wb_cur, xb_cur, g, g, q);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();

    // one k-step g (BUF = g & 1): 64 MFMAs from fragment buffer BUF.  Behind the first W4_BAR_AT of them the step's barrier: every wave has
    // read all of stage g % 4 (its fragments arrived before its MFMAs began), so step g + 4 may be requested into it; and every wave's
    // requests for step g + 1 (three steps old) have landed, so its fragments may be read into the other buffer.
    auto kstep = [&](auto FIRST, auto BUF, int stg, const char* wb, const char* xb, int ks4) __attribute__((always_inline)) {
        constexpr bool first = decltype(FIRST)::value != 0;
        constexpr int buf = decltype(BUF)::value;
        static_for<0, 64>([&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value, i = q >> 3, jj = q & 7, j = (i & 1) ? 7 - jj : jj;
            if constexpr (first) acc[i][j] = mfma16(fw[buf][i], fx[buf][j], (f32x4){0.f, 0.f, 0.f, 0.f});
            else acc[i][j] = mfma16(fw[buf][i], fx[buf][j], acc[i][j]);
            if constexpr (q == W4_BAR_AT - 1) {
                FENCE();
                if (!(W4_KO & 4)) {
                    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");      // the 16 youngest requests (steps g + 2, g + 3) may be in flight
                    __builtin_amdgcn_s_barrier();
                }
                FENCE();
            }
            // the next step's 16 fragments: ONE ds_read_b128 behind each of the 16 MFMAs after the barrier (a burst of 16 holds the wave's
            // in-order issue for as long as the LDS takes to accept 64 KB from four waves at once; the guide prices one read per MFMA gap at <= 3 cycles)
            constexpr int e = q - W4_BAR_AT;
            if constexpr (e >= 0 && e < 16) {
                if (!(W4_KO & 8)) {
                    FENCE();
                    const int ns = (stg + 1) & (NSTG - 1);
                    if constexpr (e < 8) fw[buf ^ 1][e] = *(const v8*)(smem + ns * STAGE + aw + e * 1024);
                    else fx[buf ^ 1][e - 8] = *(const v8*)(smem + ns * STAGE + ax + (e - 8) * 1024);
                    FENCE();
                }
            }
            constexpr int d = q - W4_BAR_AT - 16;
            if constexpr (d >= 0 && d % 4 == 1 && d / 4 < 4) {
                FENCE();
                if (!(W4_KO & 2)) dma_piece(wb, xb, ks4, stg, d / 4);
                FENCE();
            }
        });
        FENCE();
    };

    for (int t = 0; t < nt_mine; ++t) {
        const bool has_next = t + 1 < nt_mine;
        if (has_next) {
            nxt = coords(t + 1);
            wb_nxt = (const char*)(W + (int64_t)nxt.n0 * K);
            xb_nxt = (const char*)(A + (int64_t)nxt.m0 * K);
        }
        kstep(IC<1>{}, IC<0>{}, 0, wb_cur, xb_cur, 4);
#pragma unroll 1
        for (int g = 1; g < NK - 5; g += 2) {                  // steps 1 .. 18 request steps 5 .. 22 of this tile
            kstep(IC<0>{}, IC<1>{}, g & 3, wb_cur, xb_cur, g + 4);
            kstep(IC<0>{}, IC<0>{}, (g + 1) & 3, wb_cur, xb_cur, g + 5);
        }
        kstep(IC<0>{}, IC<1>{}, 19 & 3, wb_cur, xb_cur, 23);
        kstep(IC<0>{}, IC<0>{}, 20 & 3, wb_nxt, xb_nxt, 0);   // the last four steps request the next tile's first four (the last tile of all: its own again, unused)
        kstep(IC<0>{}, IC<1>{}, 21 & 3, wb_nxt, xb_nxt, 1);
        kstep(IC<0>{}, IC<0>{}, 22 & 3, wb_nxt, xb_nxt, 2);
        kstep(IC<0>{}, IC<1>{}, 23 & 3, wb_nxt, xb_nxt, 3);
        // ---- epilogue (serial)
        if (W4_KO & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else {
            __builtin_amdgcn_s_setreg(AVX_MODE_DX10_CLAMP_HWREG, 0);
            AVX_CLAMP_TOKEN(inva);                                 // (common.h, round 6: the clamp's multiplier is defined behind the mode write)
#pragma unroll
            for (int half = 0; half < 2; ++half) {                 // 64 features at a time: the lane-swap transpose works on four 16-feature groups
                const uint64_t a = (uint64_t)(out + (int64_t)(cur.m0 + 128 * wn) * N + cur.n0 + 128 * wm + 64 * half);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a)),
                                                                                     0, 128 * N * 2, 0x00020000);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    unsigned hA[4], hB[4];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int i = 4 * half + ii;
                        const f32x4 bq = *(const f32x4*)(bias_lane + cur.n0 + 16 * i);
                        const f32x2 y01 = gelu_erf2_h((f32x2){acc[i][j][0] + bq[0], acc[i][j][1] + bq[1]}, inva);
                        const f32x2 y23 = gelu_erf2_h((f32x2){acc[i][j][2] + bq[2], acc[i][j][3] + bq[3]}, inva);
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 1);
                        hA[ii] = __builtin_bit_cast(unsigned, (h2){(_Float16)y01[0], (_Float16)y01[1]});
                        hB[ii] = __builtin_bit_cast(unsigned, (h2){(_Float16)y23[0], (_Float16)y23[1]});
                        asm volatile("" : "+v"(hA[ii]), "+v"(hB[ii]));
                        __builtin_amdgcn_s_setreg(AVX_MODE_FP16_OVFL_HWREG, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const auto ra = __builtin_amdgcn_permlane32_swap(hA[2 * v], hA[2 * v + 1], false, false);
                        const auto rb = __builtin_amdgcn_permlane32_swap(hB[2 * v], hB[2 * v + 1], false, false);
                        const auto sa = __builtin_amdgcn_permlane16_swap(ra[0], ra[1], false, false);
                        const auto sb = __builtin_amdgcn_permlane16_swap(rb[0], rb[1], false, false);
                        typedef int i32x4_st __attribute__((ext_vector_type(4)));
                        const i32x4_st d = {(int)sa[0], (int)sb[0], (int)sa[1], (int)sb[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(d, rs, st_voff + 64 * v, j * (16 * N * 2), 2);
                    }
                }
            }
        }
        cur = nxt; wb_cur = wb_nxt; xb_cur = xb_nxt;
    }
    if (tid == 0 && clk != nullptr) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

static double run(const _Float16* A, const _Float16* W, _Float16* out, const float* bias, int M, int N, unsigned long long* clk, int iters, double* ghz) {
    CK(hipFuncSetAttribute((const void*)w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(w4_kernel, dim3(256), dim3(256), LDS_BYTES, 0, A, W, out, bias, M, N, clk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(w4_kernel, dim3(256), dim3(256), LDS_BYTES, 0, A, W, out, bias, M, N, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 512, hipMemcpyDeviceToHost));
    double s = 0;
    for (int b = 0; b < 256; ++b) s += (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0);
    *ghz = s / 256;
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const bool check = argc > 1 && !strcmp(argv[1], "check");
    const int M = check ? 8192 : 126976, N = 3072;
    const int iters = (!check && argc > 1) ? atoi(argv[1]) : 200;
    _Float16 *A, *W, *out;
    float* bias;
    unsigned long long* clk;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&clk, 512 * 8));
    std::vector<_Float16> hAm((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N);
    {
        unsigned s = 12345u;
        for (size_t i = 0; i < hAm.size(); ++i) { s = s * 1664525u + 1013904223u; hAm[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
        for (size_t i = 0; i < hW.size(); ++i) { s = s * 1664525u + 1013904223u; hW[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 1e-4f); }
        for (int i = 0; i < N; ++i) hb[i] = 0.05f * (float)(i % 17 - 8);
        CK(hipMemcpy(A, hAm.data(), hAm.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    }
    if (check) {
        CK(hipMemset(out, 0xff, (size_t)M * N * 2));
        CK(hipFuncSetAttribute((const void*)w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        hipLaunchKernelGGL(w4_kernel, dim3(256), dim3(256), LDS_BYTES, 0, A, W, out, bias, M, N, clk);
        CK(hipDeviceSynchronize());
        std::vector<_Float16> ho((size_t)M * N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        size_t bad = 0, seen = 0;
        std::vector<float> wf32((size_t)N * K);
        for (size_t i = 0; i < wf32.size(); ++i) wf32[i] = (float)hW[i];
        for (int m = 0; m < M; m += 13) {
            float a[K];
            for (int k = 0; k < K; ++k) a[k] = (float)hAm[(size_t)m * K + k];
            for (int n = 0; n < N; ++n) {
                double acc = hb[n];
                const float* w = &wf32[(size_t)n * K];
                for (int k = 0; k < K; ++k) acc += (double)a[k] * (double)w[k];
                const double ref = 0.5 * acc * (1.0 + erf(acc * 0.7071067811865476));
                const double got = (double)(float)ho[(size_t)m * N + n];
                const double err = fabs(got - ref);
                if (!(err <= 1e-3 + 1.5e-3 * fabs(ref))) { if (bad < 5) printf("  out[%d][%d] = %g, reference %g\n", m, n, got, ref); ++bad; }
                if (err > worst) worst = err;
                ++seen;
            }
        }
        printf("check: %zu outputs compared, %zu outside 1e-3 + 1.5e-3 |ref|, worst |error| %.3g  -> %s\n", seen, bad, worst, bad ? "FAILED" : "ok");
        return bad ? 1 : 0;
    }
    const double flop = 2.0 * M * N * K;
    for (int rep = 0; rep < 3; ++rep) {
        double g;
        const double us = run(A, W, out, bias, M, N, clk, iters, &g);
        printf("W4_KO %d: 4 waves x (128 x 128), one wave per SIMD: %7.1f us (%6.1f TFLOP/s, %.3f GHz)\n", W4_KO, us, flop / us / 1e6, g);
    }
    return 0;
}
