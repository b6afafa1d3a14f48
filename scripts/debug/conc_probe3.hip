// Trace the fbank corruption beside a pure-MFMA kernel to the first stage whose registers differ from the serial run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAVX_FBANK_TAPS -I include scripts/debug/conc_probe3.hip -o scripts/micro/bin/conc_probe3
#include <stdarg.h>
#include "../../avex_amd/csrc/fbank.hip"
void avexhip_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
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
// per stage: number of waves whose tap differs; first differing (wave, lane, q) with both values
__global__ void compare_taps(const float2* a, const float2* b, size_t nwaves, unsigned* counts, unsigned long long* first, float* vals) {
    const size_t w = blockIdx.x;
    for (int st = 0; st < 8; ++st) {
        const float2* pa = a + (w * 8 + st) * 512; const float2* pb = b + (w * 8 + st) * 512;
        __shared__ int bad;
        if (threadIdx.x == 0) bad = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < 512; i += blockDim.x) {
            if (__float_as_uint(pa[i].x) != __float_as_uint(pb[i].x) || __float_as_uint(pa[i].y) != __float_as_uint(pb[i].y)) {
                if (atomicAdd(&bad, 1) == 0 && atomicCAS(&first[st], 0ull, (unsigned long long)(w * 512 + i) + 1) == 0) {
                    vals[4 * st + 0] = pa[i].x; vals[4 * st + 1] = pa[i].y; vals[4 * st + 2] = pb[i].x; vals[4 * st + 3] = pb[i].y;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && bad) { atomicAdd(&counts[st], 1u); atomicAdd(&counts[8 + st], (unsigned)bad); }
        __syncthreads();
    }
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<float> win(400), mel(257 * 128, 0.f);
    for (int i = 0; i < 400; ++i) win[i] = 0.5f - 0.5f * cosf(2.f * (float)M_PI * i / 399.f);
    for (int m = 0; m < 128; ++m) for (int k = 2 * m + 1; k <= 2 * m + 3 && k < 257; ++k) mel[k * 128 + m] = 1.f - fabsf((float)(k - 2 * m - 2)) * 0.5f;
    avexhip_fbank_config fc = {400, 160, 128, 32768.f, 0.97f, 1, 1.1920929e-07f, 15.41663f, 13.11164f};
    avexhip_fbank_plan* plan = avexhip_fbank_plan_create(&fc, win.data(), mel.data());
    const int B = 8; const int64_t T = 160000; const int frames = avexhip_fbank_num_frames(plan, T);
    std::vector<float> h((size_t)B * T);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 8388608.f - 1.f) * 0.1f; }
    float *wav, *out, *sink; CK(hipMalloc(&wav, h.size() * 4)); CK(hipMemcpy(wav, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, (size_t)B * frames * 128 * 4)); CK(hipMalloc(&sink, 64));
    const size_t nwaves = (size_t)B * 125 * 4, tap_elems = nwaves * 8 * 512;
    float2 *tA, *tB; CK(hipMalloc(&tA, tap_elems * 8)); CK(hipMalloc(&tB, tap_elems * 8));
    unsigned* counts; unsigned long long* first; float* vals;
    CK(hipMalloc(&counts, 64)); CK(hipMalloc(&first, 64)); CK(hipMalloc(&vals, 128));
    auto run = [&](float2* taps, hipStream_t st) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_fbank_taps), &taps, sizeof(taps)));
        if (avexhip_fbank_forward(plan, wav, B, T, T, out, st) != 0) exit(1);
    };
    CK(hipMemset(tA, 0, tap_elems * 8));
    run(tA, sb); CK(hipDeviceSynchronize());
    const char* names[8] = {"0 windowed input", "1 after dft8 #1", "2 after twiddle #1", "3 after LDS exchange #1", "4 after dft8+twiddle #2", "5 after LDS exchange #2", "6 after dft8 #3", "7 dft8 #1 level 1 (a0..a3, b0..b3)"};
    for (int trial = 0; trial < 4; ++trial) {
        CK(hipMemset(tB, 0, tap_elems * 8)); CK(hipMemset(counts, 0, 64)); CK(hipMemset(first, 0, 64));
        CK(hipDeviceSynchronize());
        if (trial > 0) for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(aggr_mfma, dim3(1024), dim3(256), 0, sa, 2000, sink);
        run(tB, sb);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(compare_taps, dim3((unsigned)nwaves), dim3(256), 0, 0, tA, tB, nwaves, counts, first, vals);
        CK(hipDeviceSynchronize());
        unsigned hc[16]; unsigned long long hf[8]; float hv[32];
        CK(hipMemcpy(hc, counts, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv, vals, 128, hipMemcpyDeviceToHost));
        printf("== trial %d (%s), %zu waves\n", trial, trial ? "beside the MFMA kernel" : "alone", nwaves);
        for (int st : {0, 7, 1, 2, 3, 4, 5, 6}) {
            printf("   stage %-26s: %u waves differ, %u values", names[st], hc[st], hc[8 + st]);
            if (hf[st]) { const unsigned long long id = hf[st] - 1; printf("   first: wave %llu lane %llu q %llu  serial (%g, %g) now (%g, %g)", id / 512, (id % 512) / 8, id % 8, hv[4 * st], hv[4 * st + 1], hv[4 * st + 2], hv[4 * st + 3]); }
            printf("\n");
        }
    }
    return 0;
}
