// Convolutional positional embedding of the BEATs encoder as an implicit-GEMM MFMA kernel (gfx950).
//
// Restates backbone.py:52-68 (weight-normed grouped Conv1d(768, 768, k=128, pad=64, groups=16),
// SamePad drops the last output, nn.GELU) and :172-174 (x = x + pos_conv(x)).
// Per (clip, group): out[t, o] = sum_{tap<128, c<48} X[t + tap - 64, c] * W[o, tap, c], i.e. a
// [T x 6144] x [6144 x 48] product whose A-matrix is a sliding window over a (T+127) x 48 slab.
//
// One 512-thread workgroup per (clip, 512-token segment, group) -- a 10 s BEATs clip is one segment of 496 tokens, longer clips
// take several with a 64-row halo on each side -- : the group's slab (zero-padded, 96-byte rows — the
// 16x16x32 fragment reads are bank-conflict free at that stride) is staged in LDS once; each wave
// owns 64 tokens x 48 channels = 12 accumulators.  The MFMA "A" operand is the WEIGHT fragment
// (rows = output channel, read from a two-stage LDS ring that LDS-DMA fills one tap pair ahead), the "B" operand the slab
// fragment, so each lane ends with 4 consecutive channels of one token and the epilogue
// (bias, exact GELU, residual add) is float4 traffic.  K is ordered (tap, c) with c fastest: 48 is a
// multiple of 8, so an 8-element operand fragment never straddles a tap and is one 16-byte LDS read;
// the (tap, c) offsets of the three 32-wide chunks of a tap PAIR (96 = 3 x 32) are per-lane constants.
#include "common.h"

namespace {

constexpr int CG = 48;            // channels per group
constexpr int KT = 128;           // taps
constexpr int PAD = KT / 2;
constexpr int TMAX = 512;
constexpr int SLAB_ROWS = TMAX + KT;              // 640
constexpr int SLAB_BYTES = SLAB_ROWS * CG * 2;    // 61440
constexpr int KTOT = KT * CG;                     // 6144
constexpr int NU = KT / 2;                        // tap pairs: 96 = 3 x 32 k-values each
constexpr int WST = CG * 96 * 2;                  // 9216 bytes of weights per tap pair and group
constexpr int PC_LDS = SLAB_BYTES + 2 * WST;      // 79872: two workgroups per CU

// One LDS-DMA wave instruction, as inline assembly (see attention.hip: behind the builtin the compiler would wait for the
// DMA in front of every later LDS read; this kernel orders them itself with one vmcnt + barrier per tap pair).
static __device__ __forceinline__ void pc_dma16(const void* src, const char* lds_dst) {
    const unsigned lds = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char*)lds_dst;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
}

// Weights arrive packed [g][u][v][g4][o][8] (u = tap pair, v = 32-wide chunk of its 96 k-values, g4 = 8-element piece,
// o = output channel): the 9216 bytes of (g, u) are contiguous, nine LDS-DMA instructions copy them into a two-stage ring
// one tap pair ahead of the MFMAs, and the A fragment of lane (o & 15, g4) is one conflict-free 16-byte LDS read.  The
// first version of this kernel had every wave pull the group's whole 590 KB of weights from L2 into registers (8 x per
// workgroup, 19 GB per 256-clip step); now a workgroup reads them once, two workgroups share a CU and hide each other's
// barriers.
template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void posconv_kernel(const T* __restrict__ xh, const float* __restrict__ xf, const T* __restrict__ wp,
                    const float* __restrict__ bias, int Tn, int E, int G, int nseg, float* __restrict__ out, T* __restrict__ out_h) {
    extern __shared__ __attribute__((aligned(16))) char slab[];
    typedef typename Half<T>::v8 v8;
    char* wring = slab + SLAB_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x % G, bs = blockIdx.x / G;
    const int b = bs / nseg, t0 = (bs - b * nseg) * TMAX;             // this workgroup's tokens: [t0, min(t0 + 512, Tn))
    const T* xg = xh + (int64_t)b * Tn * E + g * CG;
    const char* wg = (const char*)(wp + (int64_t)g * CG * KTOT);      // this group's 64 x 9216 bytes

    auto issue_w = [&](int u) __attribute__((always_inline)) {         // nine 1 KiB pieces: one per wave, the ninth rotates
        const char* src = wg + (int64_t)u * WST;
        char* dst = wring + (u & 1) * WST;
        pc_dma16(src + wave * 1024 + lane * 16, dst + wave * 1024);
        if (wave == (u & 7)) pc_dma16(src + 8 * 1024 + lane * 16, dst + 8 * 1024);
    };
    issue_w(0);

    // ---- stage the zero-padded slab: slab row r <-> token t0 + r - 64 ---------------------------
    for (int idx = tid; idx < SLAB_ROWS * 6; idx += 512) {
        const int r = idx / 6, c = idx - r * 6;
        const int t = t0 + r - PAD;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < Tn) v = *(const uint4*)(xg + (int64_t)t * E + c * 8);
        *(uint4*)(slab + r * (CG * 2) + c * 16) = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int t16 = lane & 15, g4 = lane >> 4;
    const int tw0 = wave * 64;
    const bool has_work = t0 + tw0 < Tn;     // wave-uniform; idle waves still copy weights and join the barriers

    // per-lane constants for the three chunks of a tap pair
    int xoff[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const int off = 32 * v + 8 * g4;
        xoff[v] = (off / CG) * (CG * 2) + (off % CG) * 2;
    }
    const char* xbase = slab + (tw0 + t16) * (CG * 2);
    const int woff = (g4 * CG + t16) * 16;                              // + (v * 4 * CG + ot * 16) * 16

    f32x4 acc[3][4];
#pragma unroll
    for (int ot = 0; ot < 3; ++ot)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[ot][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int u = 0; u < NU; ++u) {
        if (u + 1 < NU) issue_w(u + 1);          // the other stage: every wave left it before the last barrier
        if (has_work) {
            const char* xu = xbase + u * (2 * CG * 2);
            const char* ws = wring + (u & 1) * WST + woff;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                v8 xfr[4], wf[3];
#pragma unroll
                for (int ot = 0; ot < 3; ++ot) wf[ot] = *(const v8*)(ws + (v * 4 * CG + ot * 16) * 16);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) xfr[tt] = *(const v8*)(xu + tt * 16 * (CG * 2) + xoff[v]);
#pragma unroll
                for (int ot = 0; ot < 3; ++ot)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[ot][tt] = mfma16(wf[ot], xfr[tt], acc[ot][tt]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // tap pair u + 1 has landed; reads of u are done
        __builtin_amdgcn_s_barrier();
    }
    if (!has_work) return;

    // ---- epilogue: + bias, exact GELU, + residual (backbone.py:68,174) --------------------------
    AVX_F16_SAT_BEGIN();      // from_hw below; set only now: with MODE.FP16_OVFL set the MFMAs above would treat a NaN operand as 0 (common.h)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int tl = tw0 + tt * 16 + t16, t = t0 + tl;
        if (t >= Tn) continue;
        const int64_t rowoff = ((int64_t)b * Tn + t) * E + g * CG;
#pragma unroll
        for (int ot = 0; ot < 3; ++ot) {
            const int n = ot * 16 + 4 * g4;
            f32x4 v = acc[ot][tt] + *(const f32x4*)(bias + g * CG + n);
            v = gelu_erf4(v);
            f32x4 r;
            if (xf) {
                r = *(const f32x4*)(xf + rowoff + n);
            } else {   // residual from the staged slab (operand precision)
                const typename Half<T>::v4 rh = *(const typename Half<T>::v4*)(slab + (tl + PAD) * (CG * 2) + n * 2);
                r = (f32x4){(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
            }
            r += v;
            if (out) *(f32x4*)(out + rowoff + n) = r;
            if (out_h) {
                typename Half<T>::v4 h;
                h[0] = Half<T>::from_hw(r[0]); h[1] = Half<T>::from_hw(r[1]);
                h[2] = Half<T>::from_hw(r[2]); h[3] = Half<T>::from_hw(r[3]);
                *(typename Half<T>::v4*)(out_h + rowoff + n) = h;
            }
        }
    }
}

// norm[k] = || v[:, :, k] ||_2 over (out, in)   (weight_norm dim=2, backbone.py:67)
__global__ __launch_bounds__(256) void posconv_norm_kernel(const float* __restrict__ v, int n_rows, int K,
                                                           float* __restrict__ norm) {
    __shared__ float red[4];
    const int k = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < n_rows; r += 256) {
        const float x = v[(int64_t)r * K + k];
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) norm[k] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

// out[g][u][v][g4][o][e] = half( v[g*cg+o][c][tap] * (gk[tap] / norm[tap]) ) with tap*cg + c = 96 u + 32 v + 8 g4 + e
// (the order posconv_kernel's LDS-DMA ring and A-fragment reads want; cg = 48)
template <typename T>
__global__ void posconv_pack_kernel(const float* __restrict__ v, const float* __restrict__ gk,
                                    const float* __restrict__ norm, int E, int cg, int K, T* __restrict__ out) {
    const int64_t total = (int64_t)E * cg * K;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg);
    const int tap = (int)((i / cg) % K);
    const int64_t o = i / ((int64_t)cg * K);  // global out channel = g*cg + o_local
    const float w = v[(o * cg + c) * K + tap] * (gk[tap] / norm[tap]);
    const int k = tap * cg + c, u = k / 96, rem = k - 96 * u, cv = rem >> 5, g4 = (rem >> 3) & 3, e = rem & 7;
    const int64_t grp = o / cg, ol = o - grp * cg;
    const int nu = cg * K / 96;
    out[(((((grp * nu + u) * 3 + cv) * 4 + g4) * cg + ol) << 3) + e] = Half<T>::from(w);
}

}  // namespace

namespace avx {

int posconv_pack(const float* g, const float* v, int E, int groups, int K, void* w_packed, int dtype,
                 hipStream_t s) {
    AVX_REQUIRE(g && v && w_packed, "posconv_pack: null argument");
    AVX_REQUIRE(groups > 0 && E % groups == 0 && K > 0, "posconv_pack: bad shape E=%d groups=%d K=%d", E, groups, K);
    const int cg = E / groups;
    float* norm = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&norm, sizeof(float) * K));
    hipLaunchKernelGGL(posconv_norm_kernel, dim3(K), dim3(256), 0, s, v, E * cg, K, norm);
    const int64_t total = (int64_t)E * cg * K;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == AVEXHIP_BF16)
        hipLaunchKernelGGL(posconv_pack_kernel<__bf16>, grid, dim3(256), 0, s, v, g, norm, E, cg, K, (__bf16*)w_packed);
    else
        hipLaunchKernelGGL(posconv_pack_kernel<_Float16>, grid, dim3(256), 0, s, v, g, norm, E, cg, K, (_Float16*)w_packed);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(norm);
    if (e != hipSuccess || e2 != hipSuccess) {
        avexhip_set_error("posconv_pack: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return AVEXHIP_ERR_HIP;
    }
    return AVEXHIP_OK;
}

int posconv(const void* x_half, const float* x_f32, const void* w_packed, const float* bias, int B, int T,
            int E, int groups, int K, float* out, void* out_half, int dtype, hipStream_t s) {
    AVX_REQUIRE(x_half && w_packed && bias && (out || out_half), "posconv: null argument");
    AVX_REQUIRE(groups > 0 && E % groups == 0 && E / groups == CG && K == KT,
                "posconv: only %d channels/group and %d taps are built (got E=%d groups=%d K=%d)", CG, KT, E, groups, K);
    AVX_REQUIRE(B > 0 && T > 0, "posconv: empty input B=%d T=%d", B, T);
    const int nseg = (T + TMAX - 1) / TMAX;
    if (dtype == AVEXHIP_BF16) {
        AVX_ENSURE_LDS(posconv_kernel<__bf16>, PC_LDS);
        hipLaunchKernelGGL(posconv_kernel<__bf16>, dim3(B * nseg * groups), dim3(512), PC_LDS, s, (const __bf16*)x_half, x_f32,
                           (const __bf16*)w_packed, bias, T, E, groups, nseg, out, (__bf16*)out_half);
    } else if (dtype == AVEXHIP_F16) {
        AVX_ENSURE_LDS(posconv_kernel<_Float16>, PC_LDS);
        hipLaunchKernelGGL(posconv_kernel<_Float16>, dim3(B * nseg * groups), dim3(512), PC_LDS, s, (const _Float16*)x_half, x_f32,
                           (const _Float16*)w_packed, bias, T, E, groups, nseg, out, (_Float16*)out_half);
    } else {
        avexhip_set_error("posconv: unknown dtype %d", dtype);
        return AVEXHIP_ERR_INVALID;
    }
    AVX_LAUNCH_CHECK();
    return AVEXHIP_OK;
}

}  // namespace avx
