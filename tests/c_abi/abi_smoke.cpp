// Plain C-ABI consumer of libavexhip.so: no Python, no torch.  Built and run by tests/test_gpu_c_abi.py on a GPU box:
//   hipcc -O1 tests/c_abi/abi_smoke.cpp -Iinclude -Lavex_amd/lib -lavexhip -Wl,-rpath,avex_amd/lib -o /tmp/abi_smoke
// Exercises what a foreign-language binding would: device buffers from hipMalloc, avexhip_cast_f32_to_half, avexhip_gemm
// (bias + GELU, fp32 and half outputs), avexhip_layernorm, avexhip_mean_pool, error reporting through avexhip_last_error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "avexhip.h"

#define CK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define AK(x) do { int rc_ = (x); if (rc_ != AVEXHIP_OK) { std::printf("avexhip error %d: %s\n", rc_, avexhip_last_error()); return 3; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main() {
    if (avexhip_abi_version() != AVEXHIP_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }
    if (avexhip_device_count() <= 0) { std::printf("no device\n"); return 1; }
    const int M = 300, N = 256, K = 192;
    unsigned seed = 7;
    std::vector<float> a(M * K), w(N * K), bias(N);
    for (auto& v : a) v = frand(seed);
    for (auto& v : w) v = 0.2f * frand(seed);
    for (auto& v : bias) v = 0.1f * frand(seed);
    float *da, *dw, *dbias, *dout, *dln, *dpool, *dg, *db;
    void *dah, *dwh, *douth;
    CK(hipMalloc(&da, sizeof(float) * M * K)); CK(hipMalloc(&dw, sizeof(float) * N * K)); CK(hipMalloc(&dbias, sizeof(float) * N));
    CK(hipMalloc(&dout, sizeof(float) * M * N)); CK(hipMalloc(&dln, sizeof(float) * M * N)); CK(hipMalloc(&dpool, sizeof(float) * N));
    CK(hipMalloc(&dg, sizeof(float) * N)); CK(hipMalloc(&db, sizeof(float) * N));
    CK(hipMalloc(&dah, 2 * M * K)); CK(hipMalloc(&dwh, 2 * N * K)); CK(hipMalloc(&douth, 2 * M * N));
    CK(hipMemcpy(da, a.data(), sizeof(float) * M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), sizeof(float) * N * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    std::vector<float> ones(N, 1.0f), zeros(N, 0.0f);
    CK(hipMemcpy(dg, ones.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, zeros.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    AK(avexhip_cast_f32_to_half(da, dah, (int64_t)M * K, AVEXHIP_F16, nullptr));
    AK(avexhip_cast_f32_to_half(dw, dwh, (int64_t)N * K, AVEXHIP_F16, nullptr));
    // the operands as the kernel sees them (rounded to f16)
    std::vector<float> ar(M * K), wr(N * K);
    AK(avexhip_cast_half_to_f32(dah, da, (int64_t)M * K, AVEXHIP_F16, nullptr));
    AK(avexhip_cast_half_to_f32(dwh, dw, (int64_t)N * K, AVEXHIP_F16, nullptr));
    CK(hipMemcpy(ar.data(), da, sizeof(float) * M * K, hipMemcpyDeviceToHost));
    CK(hipMemcpy(wr.data(), dw, sizeof(float) * N * K, hipMemcpyDeviceToHost));
    avexhip_gemm_args g;
    std::memset(&g, 0, sizeof(g));
    g.A = dah; g.lda = K; g.W = dwh; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = dbias; g.gelu = 1;
    g.out_f32 = dout; g.ldo = N; g.out_half = douth; g.ldh = N;
    AK(avexhip_gemm(&g, AVEXHIP_F16, nullptr));
    AK(avexhip_layernorm(dout, nullptr, N, dg, db, 1e-5f, M, N, dln, N, nullptr, N, AVEXHIP_F16, nullptr));
    AK(avexhip_mean_pool(dln, 1, M, N, dpool, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> out(M * N), ln(M * N), pool(N);
    CK(hipMemcpy(out.data(), dout, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ln.data(), dln, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pool.data(), dpool, sizeof(float) * N, hipMemcpyDeviceToHost));
    double err = 0, ref2 = 0, lnerr = 0, poolerr = 0;
    std::vector<double> colmean(N, 0.0);
    for (int m = 0; m < M; ++m) {
        std::vector<double> row(N);
        double mu = 0, var = 0;
        for (int n = 0; n < N; ++n) {
            double s = bias[n];
            for (int k = 0; k < K; ++k) s += (double)ar[m * K + k] * (double)wr[n * K + k];
            const double y = 0.5 * s * (1.0 + std::erf(s / std::sqrt(2.0)));
            row[n] = y; mu += y;
            err += (out[m * N + n] - y) * (out[m * N + n] - y); ref2 += y * y;
        }
        mu /= N;
        for (int n = 0; n < N; ++n) var += (row[n] - mu) * (row[n] - mu);
        var /= N;
        for (int n = 0; n < N; ++n) {
            const double z = (row[n] - mu) / std::sqrt(var + 1e-5);
            lnerr = std::fmax(lnerr, std::fabs(ln[m * N + n] - z));
            colmean[n] += z / M;
        }
    }
    for (int n = 0; n < N; ++n) poolerr = std::fmax(poolerr, std::fabs(pool[n] - colmean[n]));
    const double rel = std::sqrt(err / ref2);
    std::printf("gemm+gelu rel-L2 %.3g  layernorm max-abs %.3g  mean_pool max-abs %.3g\n", rel, lnerr, poolerr);
    // error path: an argument the library must refuse, with a message
    g.N = 100;
    const int rc = avexhip_gemm(&g, AVEXHIP_F16, nullptr);
    std::printf("bad-shape rc %d: %s\n", rc, avexhip_last_error());
    const bool ok = rel < 5e-6 && lnerr < 5e-4 && poolerr < 1e-4 && rc == AVEXHIP_ERR_INVALID;
    std::printf(ok ? "C ABI SMOKE OK\n" : "C ABI SMOKE FAILED\n");
    return ok ? 0 : 4;
}
