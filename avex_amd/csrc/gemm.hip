// MFMA GEMM for torch.nn.Linear-shaped products:  out[m,n] = epi( sum_k A[m,k] * W[n,k] ).
//
// Replaces every Linear on the BEATs path (reference call sites: beats.py:350-359 patch-embed as a
// GEMM + post_extract_proj; backbone.py:531-533 q/k/v_proj, :572 out_proj, :368 fc1, :370 fc2) with
// the bias / DeepNorm-residual / exact-erf GELU / hook-tap work fused into the epilogue.
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave),
// BK = 64, v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.  The MFMA "A" operand is the WEIGHT tile
// and the "B" operand the ACTIVATION tile, i.e. the hardware computes D[n][m]; with the 16x16 C/D
// layout (col = lane&15, row = 4*(lane>>4)+reg) every lane then owns 4 CONSECUTIVE n of one row m,
// so the epilogue moves 16-byte (fp32) / 8-byte (half) vectors and bias/residual are vector loads.
// Both operands are K-contiguous, so both are staged the same way: HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B) into a double buffer, one
// barrier per K-step, next tile's DMA in flight under the current tile's MFMAs.  LDS rows are 128 B;
// the 16-byte chunk c of row r lives in slot c ^ ((r>>1)&7) (applied on the DMA *source* address and
// on the ds_read_b128 address, never on the DMA destination) which makes every fragment read
// bank-conflict free for the 16x16x32 lane groups.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // one operand tile: 128 rows x 64 halves = 16 KiB

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <typename T, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(avx::GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Half<T>::v8 v8;
    typedef typename Half<T>::v4 v4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wid >> 1, wc = wid & 1;  // wave owns n in [n0+64wr, +64), m in [m0+64wc, +64)

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;
    const int nk = p.K / BK;

    // ---- staging -------------------------------------------------------------------------
    // LDS-DMA: wave w fills rows [32w, 32w+32) of each operand tile with 4 instructions of 8 rows.
    auto stage_dma = [&](int st, int k0) __attribute__((always_inline)) {
        char* wbase = smem + st * (2 * TILE_BYTES);
        char* abase = wbase + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rloc = wid * 32 + i * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((rloc >> 1) & 7);
            const int wrow = n0 + rloc;  // N % 128 == 0: always in range
            int arow = m0 + rloc;
            arow = arow < p.M ? arow : p.M - 1;
            const T* wsrc = W + (int64_t)wrow * p.ldw + k0 + chunk * 8;
            const T* asrc = A + (int64_t)arow * p.lda + k0 + chunk * 8;
            const int dst = (wid * 32 + i * 8) * 128;  // wave-uniform; hardware adds lane*16
            __builtin_amdgcn_global_load_lds((gptr_t*)wsrc, (lptr_t*)(wbase + dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)asrc, (lptr_t*)(abase + dst), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int st) __attribute__((always_inline)) {
        const char* wbase = smem + st * (2 * TILE_BYTES);
        const char* abase = wbase + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 wf[4], af[4];
            const int chunk = (lane >> 4) + 4 * ks;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + i * 16 + (lane & 15);
                wf[i] = *(const v8*)(wbase + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + j * 16 + (lane & 15);
                af[j] = *(const v8*)(abase + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], af[j], acc[i][j]);
        }
    };

    // ---- main loop: one barrier per K-step, next tile in flight during compute --------------
    if constexpr (GLDS) {
        stage_dma(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMA pieces of tile kt have landed
            __syncthreads();  // everyone's pieces landed; everyone finished reading buffer (kt+1)&1
            if (kt + 1 < nk) stage_dma((kt + 1) & 1, (kt + 1) * BK);
            compute(kt & 1);
        }
    } else {
        // register staging (variant 1): same LDS image, written with swizzled ds_write_b128
        uint4 rw0, rw1, rw2, rw3, ra0, ra1, ra2, ra3;
#define AVX_LD(i, k0)                                                                          \
        {                                                                                      \
            const int c = tid + 256 * i;                                                       \
            const int rloc = c >> 3, chunk = c & 7;                                            \
            int arow = m0 + rloc;                                                              \
            arow = arow < p.M ? arow : p.M - 1;                                                \
            rw##i = *(const uint4*)(W + (int64_t)(n0 + rloc) * p.ldw + (k0) + chunk * 8);      \
            ra##i = *(const uint4*)(A + (int64_t)arow * p.lda + (k0) + chunk * 8);             \
        }
#define AVX_ST(i, st)                                                                          \
        {                                                                                      \
            const int c = tid + 256 * i;                                                       \
            const int rloc = c >> 3, chunk = c & 7;                                            \
            const int off = rloc * 128 + ((chunk ^ ((rloc >> 1) & 7)) << 4);                   \
            *(uint4*)(smem + (st) * (2 * TILE_BYTES) + off) = rw##i;                           \
            *(uint4*)(smem + (st) * (2 * TILE_BYTES) + TILE_BYTES + off) = ra##i;              \
        }
        AVX_LD(0, 0) AVX_LD(1, 0) AVX_LD(2, 0) AVX_LD(3, 0)
        AVX_ST(0, 0) AVX_ST(1, 0) AVX_ST(2, 0) AVX_ST(3, 0)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            const int k1 = (kt + 1) * BK;
            if (more) { AVX_LD(0, k1) AVX_LD(1, k1) AVX_LD(2, k1) AVX_LD(3, k1) }
            compute(kt & 1);
            if (more) { AVX_ST(0, (kt + 1) & 1) AVX_ST(1, (kt + 1) & 1) AVX_ST(2, (kt + 1) & 1) AVX_ST(3, (kt + 1) & 1) }
            __syncthreads();
        }
#undef AVX_LD
#undef AVX_ST
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const float alpha = p.alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wc * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        const bool zero_row = p.row_zero != nullptr && p.row_zero[m] != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wr * 64 + i * 16 + (lane >> 4) * 4;
            f32x4 v = acc[i][j];
            if (p.bias) v += *(const f32x4*)(p.bias + n);
            if (p.out_raw) *(f32x4*)(p.out_raw + (int64_t)m * p.ldraw + n) = v;
            if (p.resid) {
                const f32x4 r = *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n);
                v = r * alpha + v;
            }
            if (p.gelu) {
                v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]);
                v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
            }
            if (zero_row) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.out_f32) *(f32x4*)(p.out_f32 + (int64_t)m * p.ldo + n) = v;
            if (p.out_half) {
                v4 h;
                h[0] = Half<T>::from(v[0]); h[1] = Half<T>::from(v[1]);
                h[2] = Half<T>::from(v[2]); h[3] = Half<T>::from(v[3]);
                *(v4*)((T*)p.out_half + (int64_t)m * p.ldh + n) = h;
            }
        }
    }
}

template <typename T>
int launch(const avx::GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    const size_t lds = 2 * 2 * TILE_BYTES;
    if (a.variant == 1) {
        hipLaunchKernelGGL((gemm_nt_kernel<T, false>), dim3(tiles), dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<T, true>), dim3(tiles), dim3(256), lds, s, a);
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace

namespace avx {

int gemm(const GemmArgs& a, int dtype, hipStream_t s) {
    AVX_REQUIRE(a.A && a.W, "gemm: A and W must be non-null");
    AVX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    AVX_REQUIRE(a.N % BN == 0, "gemm: N=%d must be a multiple of %d", a.N, BN);
    AVX_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    AVX_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
    AVX_REQUIRE(a.out_f32 || a.out_half || a.out_raw, "gemm: no output buffer");
    AVX_REQUIRE((!a.out_f32 || a.ldo % 4 == 0) && (!a.out_half || a.ldh % 4 == 0) &&
                    (!a.out_raw || a.ldraw % 4 == 0) && (!a.resid || a.ldr % 4 == 0),
                "gemm: output/residual leading dims must be multiples of 4 elements");
    if (dtype == AVEXHIP_F16) return launch<_Float16>(a, s);
    if (dtype == AVEXHIP_BF16) return launch<__bf16>(a, s);
    avexhip_set_error("gemm: unknown dtype %d", dtype);
    return AVEXHIP_ERR_INVALID;
}

}  // namespace avx
