// C-ABI surface (include/avexhip.h): error plumbing, thin wrappers over the kernel launchers and the
// BEATs encoder handle that orchestrates one forward as a fixed sequence of launches on the caller's
// stream.  Compiled with hipcc (host code only in this file).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void avexhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* avexhip_last_error(void) { return g_err; }
extern "C" int avexhip_abi_version(void) { return AVEXHIP_ABI_VERSION; }
extern "C" int avexhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------
extern "C" int avexhip_cast_f32_to_half(const float* in, void* out, int64_t n, int dtype, void* stream) {
    return avx::cast_to_half(in, out, n, dtype, (hipStream_t)stream);
}
extern "C" int avexhip_cast_half_to_f32(const void* in, float* out, int64_t n, int dtype, void* stream) {
    return avx::cast_to_f32(in, out, n, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_gemm(const avexhip_gemm_args* a, int dtype, void* stream) {
    AVX_REQUIRE(a, "gemm: null args");
    avx::GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.ln_rows = a->ln_rows; g.ln_s = a->ln_s;
    g.lnr_y = a->lnr_y; g.ldy = (int)a->ldy; g.lnr_rows = a->lnr_rows;
    g.lnr_gamma = a->lnr_gamma; g.lnr_beta = a->lnr_beta; g.stats_out = a->stats_out;
    g.ovf = a->overflow_count;
    g.A = a->A; g.lda = a->lda; g.W = a->W; g.ldw = a->ldw;
    g.M = a->M; g.N = a->N; g.K = a->K;
    g.bias = a->bias; g.resid = a->resid; g.ldr = a->ldr; g.alpha = a->alpha; g.gelu = a->gelu;
    g.resid_half = a->resid_half; g.ldrh = a->ldrh;
    g.out_f32 = a->out_f32; g.ldo = a->ldo; g.out_half = a->out_half; g.ldh = a->ldh;
    g.out_raw = a->out_raw; g.ldraw = a->ldraw; g.row_zero = nullptr; g.variant = a->variant;
    return avx::gemm(g, dtype, (hipStream_t)stream);
}
extern "C" int avexhip_ln_rowstats(const float* stats, int M, int nseg, float eps, float* rows, void* stream) {
    return avx::ln_rowstats(stats, M, nseg, eps, rows, (hipStream_t)stream);
}

namespace avx {
namespace {
std::mutex g_dev_mu;
std::map<std::pair<int, const void*>, int> g_lds_set;     // (device, kernel) -> bytes already opted in
std::map<int, int> g_cu_count;
}  // namespace

int ensure_max_dynamic_lds(const void* func, int bytes) {
    int dev = 0;
    AVX_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mu);
    auto key = std::make_pair(dev, func);
    auto it = g_lds_set.find(key);
    if (it != g_lds_set.end() && it->second >= bytes) return AVEXHIP_OK;
    AVX_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_lds_set[key] = bytes;
    return AVEXHIP_OK;
}

int device_cu_count(int* n_cu) {
    int dev = 0;
    AVX_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mu);
    auto it = g_cu_count.find(dev);
    if (it == g_cu_count.end()) {
        hipDeviceProp_t prop;
        AVX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        it = g_cu_count.emplace(dev, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256).first;
    }
    *n_cu = it->second;
    return AVEXHIP_OK;
}
}  // namespace avx

extern "C" int avexhip_layernorm(const float* in, const void* in_half, int64_t ld_in, const float* w, const float* b,
                                 float eps, int M, int C, float* out_f32, int64_t ldo, void* out_half, int64_t ldh,
                                 int dtype, void* stream) {
    return avx::layernorm(in, in_half, ld_in, w, b, eps, M, C, out_f32, ldo, out_half, ldh, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_attention(const void* qkv, int B, int T, int H, const float* bias_tab, const float* grep_w,
                                 const float* grep_b, const float* grep_a, const uint8_t* key_pad, void* out,
                                 int dtype, void* stream) {
    return avx::attention(qkv, B, T, H, bias_tab, grep_w, grep_b, grep_a, key_pad, out, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_posconv_pack(const float* g, const float* v, int E, int groups, int K, void* w_packed,
                                    int dtype, void* stream) {
    return avx::posconv_pack(g, v, E, groups, K, w_packed, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_posconv(const void* x_half, const float* x_f32, const void* w_packed, const float* bias, int B,
                               int T, int E, int groups, int K, float* out_f32, void* out_half, int dtype, void* stream) {
    return avx::posconv(x_half, x_f32, w_packed, bias, B, T, E, groups, K, out_f32, out_half, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_token_embed_ln(const void* patches_half, const float* pos, const float* cls, const float* ln_w, const float* ln_b, float eps,
                                      int B, int n_patches, int C, void* out_half, float* out_f32, int dtype, void* stream) {
    return avx::token_embed_ln(patches_half, pos, cls, ln_w, ln_b, eps, B, n_patches, C, out_half, out_f32, dtype, (hipStream_t)stream);
}

extern "C" int avexhip_mean_pool(const float* in, int B, int T, int C, float* out, void* stream) {
    return avx::mean_pool(in, B, T, C, nullptr, out, (hipStream_t)stream);
}

// T5 bidirectional bucket, fp32 arithmetic exactly as the reference writes it (backbone.py:438-473):
//   nb = num_buckets/2; out = (rel > 0) * nb; a = |rel|; max_exact = nb/2;
//   large = max_exact + trunc( log(float(a)/max_exact) / log(max_distance/max_exact) * (nb - max_exact) )
extern "C" int avexhip_rel_bucket(int rel, int num_buckets, int max_distance) {
    const int nb = num_buckets / 2;
    int out = rel > 0 ? nb : 0;
    const int a = rel < 0 ? -rel : rel;
    const int max_exact = nb / 2;
    if (a < max_exact) return out + a;
    const float num = logf((float)a / (float)max_exact);
    const float den = (float)log((double)max_distance / (double)max_exact);
    const float val = num / den * (float)(nb - max_exact);
    int big = max_exact + (int)val;
    if (big > nb - 1) big = nb - 1;
    return out + big;
}

// ---------------------------------------------------------------------------------------------
// BEATs handle
// ---------------------------------------------------------------------------------------------
const avx::FbankDev* avexhip_fbank_plan_dev(const avexhip_fbank_plan* plan);

namespace {

struct Layer {
    void* w_qkv = nullptr; float* b_qkv = nullptr;
    void* w_o = nullptr;   float* b_o = nullptr;
    float* grep_w = nullptr; float* grep_b = nullptr; float* grep_a = nullptr;
    float* ln1_w = nullptr; float* ln1_b = nullptr;
    void* w_fc1 = nullptr; float* b_fc1 = nullptr;
    void* w_fc2 = nullptr; float* b_fc2 = nullptr;
    float* ln2_w = nullptr; float* ln2_b = nullptr;
    // LayerNorm-folded copies (see GemmArgs): fc1 consumes LN1 of this layer, QKV consumes LN2 of the previous layer
    void* w_fc1_f = nullptr; float* b_fc1_f = nullptr; float* s_fc1 = nullptr;
    void* w_qkv_f = nullptr; float* b_qkv_f = nullptr; float* s_qkv = nullptr;
    // residual-side folds (GemmArgs::lnr_prefolded): fc2 adds alpha * LN1(y1) of this layer, out_proj alpha * LN2(y2) of the previous layer;
    // ga = alpha * gamma, bb = bias + alpha * beta
    float* ga_fc2 = nullptr; float* bb_fc2 = nullptr;
    float* ga_o = nullptr; float* bb_o = nullptr;
};

struct StageRec {
    std::string name;
    double flops;
    hipEvent_t e0, e1;
};

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// window / mel bank as the reference builds them (beats.py:75,82-118), fp32 arithmetic
void default_window(int win, std::vector<float>& w) {
    w.resize(win);
    for (int n = 0; n < win; ++n) {
        // torch.hann_window(periodic=False) op for op in fp32
        const float hann = cosf((float)n * (float)(M_PI * 2.0 / (double)(win - 1))) * -0.5f + 0.5f;
        w[n] = powf(hann, 0.85f);
    }
}
void default_mel(int n_fft, int n_mels, float sr, float low, float high, std::vector<float>& fb) {
    const int nb = n_fft / 2;
    const float bin_w = sr / (float)n_fft;
    const float mel_low = (float)(1127.0 * log(1.0 + (double)low / 700.0));
    const float mel_high = (float)(1127.0 * log(1.0 + (double)high / 700.0));
    const float delta = (float)(((double)mel_high - (double)mel_low) / (double)(n_mels + 1));
    fb.assign((size_t)(nb + 1) * n_mels, 0.f);
    for (int m = 0; m < n_mels; ++m) {
        const float left = mel_low + (float)m * delta;
        const float center = mel_low + ((float)m + 1.0f) * delta;
        const float right = mel_low + ((float)m + 2.0f) * delta;
        for (int k = 0; k < nb; ++k) {
            const float f = bin_w * (float)k;
            const float mel = 1127.0f * logf(1.0f + f / 700.0f);
            const float up = (mel - left) / (center - left);
            const float down = (right - mel) / (right - center);
            const float v = fmaxf(0.f, fminf(up, down));
            fb[(size_t)k * n_mels + m] = v;
        }
    }
}

}  // namespace

struct avexhip_beats {
    avexhip_beats_config cfg;
    int dtype = AVEXHIP_F16;
    int E = 0, F = 0, H = 0, L = 0, D = 0, P = 0, NM = 0, chunk = 256;
    bool fast = false;   // residual stream / pre-LN sums in the operand type
    bool ln_fold = false;  // fast mode: LayerNorms between the GEMMs folded into their epilogues
    int nstreams = 1;    // chunks of one forward run concurrently on this many streams (caller's + side streams)
    bool fe_in_lane = true;    // frontend of a chunk on the chunk's own lane stream (AVEX_AMD_FRONTEND_IN_LANE=0: all frontends before the fork)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    float alpha = 1.f;
    avexhip_fbank_plan* fb = nullptr;
    void* w_patch = nullptr;
    float* ln0_w = nullptr; float* ln0_b = nullptr;
    void* w_post = nullptr; float* b_post = nullptr;
    void* w_pc = nullptr; float* b_pc = nullptr;
    float* lnE_w = nullptr; float* lnE_b = nullptr;
    std::vector<Layer> layers;
    std::vector<float> rel_table;  // host [num_buckets, H]; empty if no relative position embedding
    std::map<int, float*> bias_tabs;       // per token count T: [H, 2T-1] Toeplitz rows; at most BIAS_TAB_CACHE entries, least recently used evicted
    std::vector<int> bias_tab_lru;         // token counts, most recent last
    std::vector<void*> allocs;
    // range alarm of the f16 conversions (GemmArgs::ovf): device counter every GEMM of a forward adds to, mirrored to pinned host
    // memory by an asynchronous copy at the end of each forward (read without a synchronisation by avexhip_beats_overflow_count)
    unsigned int* d_ovf = nullptr;
    unsigned int* h_ovf = nullptr;
    bool profiling = false;
    std::vector<StageRec> recs;
    std::vector<std::string> prof_names;
    std::vector<const char*> prof_name_ptrs;
    std::vector<float> prof_ms;
    std::vector<double> prof_flops;

    ~avexhip_beats() {
        for (void* p : allocs) (void)hipFree(p);
        for (auto& kv : bias_tabs) (void)hipFree(kv.second);
        if (d_ovf) (void)hipFree(d_ovf);
        if (h_ovf) (void)hipHostFree(h_ovf);
        if (fb) avexhip_fbank_plan_destroy(fb);
        for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        for (int i = 0; i < 3; ++i) { if (side[i]) (void)hipStreamDestroy(side[i]); if (ev_join[i]) (void)hipEventDestroy(ev_join[i]); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

namespace {

struct Table {
    const avexhip_tensor* t;
    int n;
    const avexhip_tensor* find(const std::string& name) const {
        for (int i = 0; i < n; ++i) {
            if (!t[i].name) continue;
            const char* nm = t[i].name;
            if (strncmp(nm, "backbone.", 9) == 0) nm += 9;
            if (name == nm) return &t[i];
        }
        return nullptr;
    }
};

// copy an fp32 tensor to the device (source may be host or device memory)
int dev_f32(avexhip_beats* h, const Table& tb, const std::string& name, int64_t numel, float** out, bool required = true) {
    const avexhip_tensor* t = tb.find(name);
    if (!t) {
        if (!required) { *out = nullptr; return AVEXHIP_OK; }
        avexhip_set_error("beats_create: tensor '%s' missing from the weight table", name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    if (t->numel != numel || !t->data) {
        avexhip_set_error("beats_create: tensor '%s' has %lld elements, expected %lld", name.c_str(), (long long)t->numel, (long long)numel);
        return AVEXHIP_ERR_INVALID;
    }
    float* d = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * (size_t)numel));
    h->allocs.push_back(d);
    AVX_HIP_CHECK(hipMemcpy(d, t->data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
    *out = d;
    return AVEXHIP_OK;
}

// fp32 tensor -> half copy at dst (device), via a temporary fp32 device staging buffer
int dev_half_into(avexhip_beats* h, const Table& tb, const std::string& name, int64_t numel, void* dst) {
    const avexhip_tensor* t = tb.find(name);
    if (!t) {
        avexhip_set_error("beats_create: tensor '%s' missing from the weight table", name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    if (t->numel != numel || !t->data) {
        avexhip_set_error("beats_create: tensor '%s' has %lld elements, expected %lld", name.c_str(), (long long)t->numel, (long long)numel);
        return AVEXHIP_ERR_INVALID;
    }
    float* tmp = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(float) * (size_t)numel));
    hipError_t e = hipMemcpy(tmp, t->data, sizeof(float) * (size_t)numel, hipMemcpyDefault);
    int rc = AVEXHIP_OK;
    if (e != hipSuccess) {
        avexhip_set_error("beats_create: copy of '%s' failed: %s", name.c_str(), hipGetErrorString(e));
        rc = AVEXHIP_ERR_HIP;
    } else {
        rc = avx::cast_to_half(tmp, dst, numel, h->dtype, nullptr);
        if (rc == AVEXHIP_OK && hipDeviceSynchronize() != hipSuccess) {
            avexhip_set_error("beats_create: cast of '%s' failed", name.c_str());
            rc = AVEXHIP_ERR_HIP;
        }
    }
    (void)hipFree(tmp);
    return rc;
}

int dev_half(avexhip_beats* h, const Table& tb, const std::string& name, int64_t numel, void** out) {
    void* d = nullptr;
    AVX_HIP_CHECK(hipMalloc(&d, 2 * (size_t)numel));
    h->allocs.push_back(d);
    *out = d;
    return dev_half_into(h, tb, name, numel, d);
}

// W' = half(W * diag(gamma)), b' = b + W beta, s[n] = sum_k float(W'[n][k]) for a consumer of LayerNorm(y; gamma, beta)
// (W: [N, K] fp32 host rows gathered from the table by the caller)
int fold_ln(avexhip_beats* h, const std::vector<float>& W, const std::vector<float>& b, int N, int K, const float* gamma_dev,
            const float* beta_dev, void** w_out, float** b_out, float** s_out) {
    std::vector<float> gamma(K), beta(K), Wg((size_t)N * K), bf(N);
    AVX_HIP_CHECK(hipMemcpy(gamma.data(), gamma_dev, sizeof(float) * K, hipMemcpyDefault));
    AVX_HIP_CHECK(hipMemcpy(beta.data(), beta_dev, sizeof(float) * K, hipMemcpyDefault));
    for (int n = 0; n < N; ++n) {
        double acc = b[n];
        const float* wr = &W[(size_t)n * K];
        float* wo = &Wg[(size_t)n * K];
        for (int k = 0; k < K; ++k) { wo[k] = wr[k] * gamma[k]; acc += (double)wr[k] * (double)beta[k]; }
        bf[n] = (float)acc;
    }
    float* tmp = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(float) * (size_t)N * K));
    void* wd = nullptr; float* bd = nullptr; float* sd = nullptr;
    int rc = AVEXHIP_OK;
    if (hipMalloc(&wd, 2 * (size_t)N * K) != hipSuccess || hipMalloc((void**)&bd, sizeof(float) * N) != hipSuccess ||
        hipMalloc((void**)&sd, sizeof(float) * N) != hipSuccess) {
        avexhip_set_error("beats_create: device allocation for folded weights failed");
        rc = AVEXHIP_ERR_HIP;
    }
    if (wd) h->allocs.push_back(wd);
    if (bd) h->allocs.push_back(bd);
    if (sd) h->allocs.push_back(sd);
    if (rc == AVEXHIP_OK && (hipMemcpy(tmp, Wg.data(), sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice) != hipSuccess ||
                             hipMemcpy(bd, bf.data(), sizeof(float) * N, hipMemcpyHostToDevice) != hipSuccess)) {
        avexhip_set_error("beats_create: upload of folded weights failed");
        rc = AVEXHIP_ERR_HIP;
    }
    if (rc == AVEXHIP_OK) rc = avx::cast_to_half(tmp, wd, (int64_t)N * K, h->dtype, nullptr);
    if (rc == AVEXHIP_OK) rc = avx::row_sum_half(wd, N, K, sd, h->dtype, nullptr);
    if (rc == AVEXHIP_OK && hipDeviceSynchronize() != hipSuccess) { avexhip_set_error("beats_create: folding failed"); rc = AVEXHIP_ERR_HIP; }
    (void)hipFree(tmp);
    *w_out = wd; *b_out = bd; *s_out = sd;
    return rc;
}

// gather an fp32 tensor of the table into a host vector (appending)
int host_f32(const Table& tb, const std::string& name, int64_t numel, std::vector<float>& out) {
    const avexhip_tensor* t = tb.find(name);
    if (!t || t->numel != numel || !t->data) {
        avexhip_set_error("beats_create: tensor '%s' missing or mis-sized", name.c_str());
        return AVEXHIP_ERR_MISSING;
    }
    const size_t o = out.size();
    out.resize(o + (size_t)numel);
    AVX_HIP_CHECK(hipMemcpy(out.data() + o, t->data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
    return AVEXHIP_OK;
}

int build(avexhip_beats* h, const avexhip_tensor* tensors, int n) {
    const avexhip_beats_config& c = h->cfg;
    const Table tb{tensors, n};
    const int E = h->E, F = h->F, H = h->H, D = h->D, P = h->P;
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)

    // frontend plan: the checkpoint's persistent buffers when present, else the reference's formulas
    {
        const int win = (int)(c.sample_frequency * c.frame_length_ms / 1000.0f);
        const int hop = (int)(c.sample_frequency * c.frame_shift_ms / 1000.0f);
        AVX_REQUIRE(win > 0 && win <= 512, "beats_create: frame length %d samples unsupported (n_fft fixed at 512)", win);
        std::vector<float> hw, hm;
        const avexhip_tensor* tw = tb.find("fbank.window");
        const avexhip_tensor* tm = tb.find("fbank.mel_fb");
        if (tw && tw->numel == win && tw->data) {
            hw.resize(win);
            AVX_HIP_CHECK(hipMemcpy(hw.data(), tw->data, sizeof(float) * win, hipMemcpyDefault));
        } else {
            default_window(win, hw);
        }
        if (tm && tm->numel == (int64_t)257 * c.num_mel_bins && tm->data) {
            hm.resize((size_t)257 * c.num_mel_bins);
            AVX_HIP_CHECK(hipMemcpy(hm.data(), tm->data, sizeof(float) * hm.size(), hipMemcpyDefault));
        } else {
            default_mel(512, c.num_mel_bins, c.sample_frequency, 20.0f, c.sample_frequency / 2.0f, hm);
        }
        avexhip_fbank_config fc;
        fc.win_length = win; fc.hop_length = hop; fc.n_mels = c.num_mel_bins;
        fc.input_scale = 32768.0f; fc.preemph = 0.97f; fc.remove_dc = 1; fc.log_floor = 1.1920929e-07f;
        fc.norm_mean = c.fbank_mean; fc.norm_div = 2.0f * c.fbank_std;
        h->fb = avexhip_fbank_plan_create(&fc, hw.data(), hm.data());
        if (!h->fb) return AVEXHIP_ERR_HIP;
    }

    RC(dev_half(h, tb, "patch_embedding.weight", (int64_t)D * P * P, &h->w_patch));
    RC(dev_f32(h, tb, "layer_norm.weight", D, &h->ln0_w));
    RC(dev_f32(h, tb, "layer_norm.bias", D, &h->ln0_b));
    if (D != E || tb.find("post_extract_proj.weight")) {
        RC(dev_half(h, tb, "post_extract_proj.weight", (int64_t)E * D, &h->w_post));
        RC(dev_f32(h, tb, "post_extract_proj.bias", E, &h->b_post));
    }
    // positional conv: fold weight-norm and repack
    {
        const int K = c.conv_pos, G = c.conv_pos_groups, cg = E / G;
        float *g = nullptr, *v = nullptr;
        RC(dev_f32(h, tb, "encoder.pos_conv.0.parametrizations.weight.original0", K, &g));
        RC(dev_f32(h, tb, "encoder.pos_conv.0.parametrizations.weight.original1", (int64_t)E * cg * K, &v));
        AVX_HIP_CHECK(hipMalloc(&h->w_pc, 2 * (size_t)E * cg * K));
        h->allocs.push_back(h->w_pc);
        RC(avx::posconv_pack(g, v, E, G, K, h->w_pc, h->dtype, nullptr));
        RC(dev_f32(h, tb, "encoder.pos_conv.0.bias", E, &h->b_pc));
    }
    RC(dev_f32(h, tb, "encoder.layer_norm.weight", E, &h->lnE_w));
    RC(dev_f32(h, tb, "encoder.layer_norm.bias", E, &h->lnE_b));

    h->layers.resize(h->L);
    for (int i = 0; i < h->L; ++i) {
        Layer& ly = h->layers[i];
        const std::string p = "encoder.layers." + std::to_string(i) + ".";
        const std::string sa = p + "self_attn.";
        // fused QKV weight [3E, E] and bias [3E]
        AVX_HIP_CHECK(hipMalloc(&ly.w_qkv, 2 * (size_t)3 * E * E));
        h->allocs.push_back(ly.w_qkv);
        RC(dev_half_into(h, tb, sa + "q_proj.weight", (int64_t)E * E, ly.w_qkv));
        RC(dev_half_into(h, tb, sa + "k_proj.weight", (int64_t)E * E, (char*)ly.w_qkv + 2 * (size_t)E * E));
        RC(dev_half_into(h, tb, sa + "v_proj.weight", (int64_t)E * E, (char*)ly.w_qkv + 4 * (size_t)E * E));
        AVX_HIP_CHECK(hipMalloc((void**)&ly.b_qkv, sizeof(float) * 3 * E));
        h->allocs.push_back(ly.b_qkv);
        const char* bn[3] = {"q_proj.bias", "k_proj.bias", "v_proj.bias"};
        for (int j = 0; j < 3; ++j) {
            const avexhip_tensor* t = tb.find(sa + bn[j]);
            if (!t || t->numel != E) {
                avexhip_set_error("beats_create: tensor '%s%s' missing or mis-sized", sa.c_str(), bn[j]);
                return AVEXHIP_ERR_MISSING;
            }
            AVX_HIP_CHECK(hipMemcpy(ly.b_qkv + (size_t)j * E, t->data, sizeof(float) * E, hipMemcpyDefault));
        }
        RC(dev_half(h, tb, sa + "out_proj.weight", (int64_t)E * E, &ly.w_o));
        RC(dev_f32(h, tb, sa + "out_proj.bias", E, &ly.b_o));
        if (c.gru_rel_pos) {
            RC(dev_f32(h, tb, sa + "grep_linear.weight", 8 * (E / H), &ly.grep_w));
            RC(dev_f32(h, tb, sa + "grep_linear.bias", 8, &ly.grep_b));
            RC(dev_f32(h, tb, sa + "grep_a", H, &ly.grep_a));
        }
        RC(dev_f32(h, tb, p + "self_attn_layer_norm.weight", E, &ly.ln1_w));
        RC(dev_f32(h, tb, p + "self_attn_layer_norm.bias", E, &ly.ln1_b));
        RC(dev_half(h, tb, p + "fc1.weight", (int64_t)F * E, &ly.w_fc1));
        RC(dev_f32(h, tb, p + "fc1.bias", F, &ly.b_fc1));
        RC(dev_half(h, tb, p + "fc2.weight", (int64_t)E * F, &ly.w_fc2));
        RC(dev_f32(h, tb, p + "fc2.bias", E, &ly.b_fc2));
        RC(dev_f32(h, tb, p + "final_layer_norm.weight", E, &ly.ln2_w));
        RC(dev_f32(h, tb, p + "final_layer_norm.bias", E, &ly.ln2_b));
        if (h->ln_fold) {
            auto two = [&](float** ga, float** bb) -> int {
                AVX_HIP_CHECK(hipMalloc((void**)ga, sizeof(float) * 2 * (size_t)E));
                h->allocs.push_back(*ga);
                *bb = *ga + E;
                return AVEXHIP_OK;
            };
            RC(two(&ly.ga_fc2, &ly.bb_fc2));
            RC(avx::lnr_fold(ly.ln1_w, ly.ln1_b, ly.b_fc2, h->alpha, E, ly.ga_fc2, ly.bb_fc2, nullptr));
            if (i > 0) {
                const Layer& prev = h->layers[i - 1];
                RC(two(&ly.ga_o, &ly.bb_o));
                RC(avx::lnr_fold(prev.ln2_w, prev.ln2_b, ly.b_o, h->alpha, E, ly.ga_o, ly.bb_o, nullptr));
            }
            std::vector<float> Wh, bh;
            RC(host_f32(tb, p + "fc1.weight", (int64_t)F * E, Wh));
            RC(host_f32(tb, p + "fc1.bias", F, bh));
            RC(fold_ln(h, Wh, bh, F, E, ly.ln1_w, ly.ln1_b, &ly.w_fc1_f, &ly.b_fc1_f, &ly.s_fc1));
            if (i > 0) {   // QKV of layer i reads LN2 of layer i-1
                Wh.clear(); bh.clear();
                RC(host_f32(tb, sa + "q_proj.weight", (int64_t)E * E, Wh));
                RC(host_f32(tb, sa + "k_proj.weight", (int64_t)E * E, Wh));
                RC(host_f32(tb, sa + "v_proj.weight", (int64_t)E * E, Wh));
                RC(host_f32(tb, sa + "q_proj.bias", E, bh));
                RC(host_f32(tb, sa + "k_proj.bias", E, bh));
                RC(host_f32(tb, sa + "v_proj.bias", E, bh));
                const Layer& prev = h->layers[i - 1];
                RC(fold_ln(h, Wh, bh, 3 * E, E, prev.ln2_w, prev.ln2_b, &ly.w_qkv_f, &ly.b_qkv_f, &ly.s_qkv));
            }
        }
    }
    // shared relative-position table (owned by layer 0, backbone.py:100-103)
    if (c.num_buckets > 0) {
        const avexhip_tensor* t = tb.find("encoder.layers.0.self_attn.relative_attention_bias.weight");
        if (!t || t->numel != (int64_t)c.num_buckets * H) {
            avexhip_set_error("beats_create: relative_attention_bias.weight missing or mis-sized");
            return AVEXHIP_ERR_MISSING;
        }
        h->rel_table.resize((size_t)c.num_buckets * H);
        AVX_HIP_CHECK(hipMemcpy(h->rel_table.data(), t->data, sizeof(float) * h->rel_table.size(), hipMemcpyDefault));
    }
#undef RC
    AVX_HIP_CHECK(hipDeviceSynchronize());
    return AVEXHIP_OK;
}

// [H, 2T-1] Toeplitz rows of compute_bias (backbone.py:475-492), cached per T
int bias_tab_for(avexhip_beats* h, int T, float** out) {
    *out = nullptr;
    if (h->rel_table.empty()) return AVEXHIP_OK;
    auto touch = [&](int t) {
        auto& v = h->bias_tab_lru;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == t) { v.erase(v.begin() + (long)i); break; }
        v.push_back(t);
    };
    auto it = h->bias_tabs.find(T);
    if (it != h->bias_tabs.end()) { touch(T); *out = it->second; return AVEXHIP_OK; }
    // variable-length inference meets a new T per clip length: the cache is bounded (3 MB per entry near T = 32768).  hipFree waits for
    // the device, so a table still read by a kernel in flight is never pulled from under it.
    constexpr size_t BIAS_TAB_CACHE = 16;
    while (h->bias_tabs.size() >= BIAS_TAB_CACHE && !h->bias_tab_lru.empty()) {
        const int victim = h->bias_tab_lru.front();
        h->bias_tab_lru.erase(h->bias_tab_lru.begin());
        auto vit = h->bias_tabs.find(victim);
        if (vit != h->bias_tabs.end()) { (void)hipFree(vit->second); h->bias_tabs.erase(vit); }
    }
    const int H = h->H, W = 2 * T - 1;
    std::vector<float> host((size_t)H * W);
    for (int r = 0; r < W; ++r) {
        const int bucket = avexhip_rel_bucket(r - (T - 1), h->cfg.num_buckets, h->cfg.max_distance);
        for (int hh = 0; hh < H; ++hh) host[(size_t)hh * W + r] = h->rel_table[(size_t)bucket * H + hh];
    }
    float* d = nullptr;
    AVX_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * host.size()));
    AVX_HIP_CHECK(hipMemcpy(d, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice));
    h->bias_tabs[T] = d;
    touch(T);
    *out = d;
    return AVEXHIP_OK;
}

struct Ws {
    char* patches; float* f0; char* h0; float* x; char* xh; float* pre; char* preh; char* qkv; char* ah; char* hh; float* raw;
    float* st1; float* st2;   // folded LayerNorm: per-row partial statistics [M][E/64][2] of y1 (preh) and y2 (xh)
    float* r1; float* r2;     // ... reduced to (rstd, -mu rstd) per row by avx::ln_rowstats
    size_t total;
};

Ws carve(const avexhip_beats* h, char* base, int Bc, int Tt) {
    const size_t M = (size_t)Bc * Tt;
    const size_t PP = (size_t)h->P * h->P;
    Ws w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    w.patches = take(M * PP * 2);
    w.f0 = (float*)take(M * h->D * 4);
    w.h0 = take(M * h->D * 2);
    w.x = (float*)take(M * h->E * 4);
    w.xh = take(M * h->E * 2);
    w.pre = (float*)take(h->fast ? 256 : M * h->E * 4);
    w.preh = take(h->fast ? M * h->E * 2 : 256);
    w.qkv = take(M * 3 * h->E * 2);
    w.ah = take(M * h->E * 2);
    w.hh = take(M * h->F * 2);
    w.raw = (float*)take(M * h->E * 4);
    w.st1 = (float*)take(h->ln_fold ? M * (h->E / 64) * 8 : 256);
    w.st2 = (float*)take(h->ln_fold ? M * (h->E / 64) * 8 : 256);
    w.r1 = (float*)take(h->ln_fold ? (M + 256) * 8 : 256);
    w.r2 = (float*)take(h->ln_fold ? (M + 256) * 8 : 256);
    w.total = off;
    return w;
}

// how a batch is split: chunk size and number of concurrent lanes (streams)
void plan_chunks(const avexhip_beats* h, int B, int Tt, int* chunk, int* lanes) {
    // max_chunk_clips is sized for 10 s clips (<= 512 tokens); longer clips keep the same number of TOKEN rows per pass
    int cap = h->chunk;
    if (Tt > 512) { cap = (int)(((int64_t)h->chunk * 512) / Tt); if (cap < 1) cap = 1; }
    int c = B < cap ? B : cap;
    int l = 1;
    if (h->nstreams > 1 && B > 1) {
        const int per = (B + h->nstreams - 1) / h->nstreams;
        if (per < c) c = per;
        l = (B + c - 1) / c;
        if (l > h->nstreams) l = h->nstreams;
    }
    *chunk = c; *lanes = l;
}

struct Prof {
    avexhip_beats* h;
    hipStream_t s;
    size_t next = 0;
    void begin(const char* name, double flops) {
        if (!h->profiling) return;
        if (next == h->recs.size()) {
            StageRec r;
            (void)hipEventCreate(&r.e0);
            (void)hipEventCreate(&r.e1);
            h->recs.push_back(r);
        }
        h->recs[next].name = name;
        h->recs[next].flops = flops;
        (void)hipEventRecord(h->recs[next].e0, s);
    }
    void end() {
        if (!h->profiling) return;
        (void)hipEventRecord(h->recs[next].e1, s);
        ++next;
    }
};

int forward_impl(avexhip_beats* h, const float* wav, const float* fbank_in, int B, int64_t T, int64_t stride, int frames,
                 const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out, int hook_pooled,
                 float* features_out, float* pooled_out, void* workspace, size_t ws_bytes, hipStream_t s) {
    const int E = h->E, F = h->F, H = h->H, D = h->D, P = h->P, L = h->L, NM = h->NM, dt = h->dtype;
    const int nt = frames / P, nf = NM / P;
    const int Tt = nt * nf;
    AVX_REQUIRE(Tt >= 1, "beats_forward: input too short (%d frames -> 0 tokens)", frames);
    AVX_REQUIRE(hook_mask == 0 || hook_out, "beats_forward: hook_mask set but hook_out is NULL");
    AVX_REQUIRE((hook_mask >> (L + 1)) == 0, "beats_forward: hook_mask has bits beyond layer %d", L);
    // hook 0 is post_extract_proj's output; a model with embed_dim == encoder_embed_dim has no such layer (beats.py:357-358)
    AVX_REQUIRE(!(hook_mask & 1u) || h->w_post, "beats_forward: hook 0 (post_extract_proj) requested but this model has no post_extract_proj");
    for (int i = 0; i <= L; ++i)
        AVX_REQUIRE(!((hook_mask >> i) & 1u) || hook_out[i], "beats_forward: hook %d selected but hook_out[%d] is NULL", i, i);
    int chunk = 1, lanes = 1;
    plan_chunks(h, B, Tt, &chunk, &lanes);
    const Ws need = carve(h, nullptr, chunk, Tt);
    if (!workspace || ws_bytes < need.total * (size_t)lanes) {
        avexhip_set_error("beats_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, need.total * (size_t)lanes);
        return AVEXHIP_ERR_WORKSPACE;
    }
    if (h->profiling) lanes = 1;   // per-kernel event timing needs the kernels alone on the device
    float* bias_tab = nullptr;
    int rc = bias_tab_for(h, Tt, &bias_tab);
    if (rc != AVEXHIP_OK) return rc;
    const avx::FbankDev* fbd = avexhip_fbank_plan_dev(h->fb);
    Prof prof{h, s};
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)

    // Chunks are processed in rounds of `lanes`; each chunk's whole pipeline, frontend included, runs on its lane's stream.
    // (Round 1 kept the frontends in front of the fork because the FFT kernel computed wrong values beside another lane's GEMMs.
    // Root cause, found in round 2: the compiler had vectorised the complex butterflies into v_pk_add_f32 / v_pk_mul_f32 with
    // op_sel:[0,1], a form that reads a wrong operand on gfx950 while another wave on the CU issues MFMAs -- see
    // avex_amd/isa_lint.py, which now keeps that form out of the whole library.  AVEX_AMD_FRONTEND_IN_LANE=0 restores the old order.)
    auto frontend = [&](int c0, int Bc, const Ws& w, hipStream_t fs) -> int {
        // 1. frontend -> patch-major half tokens [M, P*P]
        const double Md = (double)Bc * Tt;
        if (wav) {
            prof.begin("fbank", Md / Tt * frames * (5.0 * 512 * 9 + 2.0 * 504));
            RC(avx::fbank(*fbd, wav + (size_t)c0 * stride, Bc, T, stride, frames, nullptr, w.patches, P, dt, fs));
            prof.end();
        } else {
            prof.begin("patchify", 0.0);
            RC(avx::patchify(fbank_in + (size_t)c0 * frames * NM, Bc, frames, NM, P, w.patches, dt, fs));
            prof.end();
        }
        return AVEXHIP_OK;
    };
    for (int r0 = 0; r0 < B; r0 += chunk * lanes) {
    if (!h->fe_in_lane) {
        for (int li = 0; li < lanes; ++li) {
            const int c0 = r0 + li * chunk;
            if (c0 >= B) break;
            const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
            RC(frontend(c0, Bc, carve(h, (char*)workspace + (size_t)li * need.total, chunk, Tt), s));
        }
    }
    if (lanes > 1) {
        // fresh events every round: a re-recorded event must never be observed by a wait enqueued earlier
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        AVX_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        AVX_HIP_CHECK(hipEventRecord(h->ev_fork, s));
        for (int i = 0; i + 1 < lanes; ++i) AVX_HIP_CHECK(hipStreamWaitEvent(h->side[i], h->ev_fork, 0));
    }
    for (int li = 0; li < lanes; ++li) {
        const int c0 = r0 + li * chunk;
        if (c0 >= B) break;
        const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
        const int M = Bc * Tt;
        const int lane_id = li;
        hipStream_t cs = lane_id == 0 ? s : h->side[lane_id - 1];
        const Ws w = carve(h, (char*)workspace + (size_t)lane_id * need.total, chunk, Tt);
        const uint8_t* pad = frame_pad ? frame_pad + (size_t)c0 * Tt : nullptr;
        const double Md = (double)M;
        if (h->fe_in_lane) RC(frontend(c0, Bc, w, cs));

        // 2. patch embedding (Conv2d as GEMM) -> LayerNorm(D) -> post_extract_proj
        // "fast" keeps the residual stream (post-LN x) and the pre-LN sums in the operand type between
        // kernels; otherwise they are fp32.  x32 / pre32 / preh below are NULL when unused.
        const bool fast = h->fast;
        float* x32 = w.x;                       // fp32 x (precise mode; in fast mode only hook 0 / final output scratch)
        float* pre32 = fast ? nullptr : w.pre;
        void* preh = fast ? w.preh : nullptr;
        const bool hook0 = (hook_mask & 1u) != 0;
        avx::GemmArgs g;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.patches; g.lda = P * P; g.W = h->w_patch; g.ldw = P * P; g.M = M; g.N = D; g.K = P * P;
        if (fast) { g.out_half = w.h0; g.ldh = D; } else { g.out_f32 = w.f0; g.ldo = D; }
        prof.begin("gemm.patch_embed", 2.0 * Md * D * P * P);
        RC(avx::gemm(g, dt, cs));
        prof.end();
        prof.begin("layernorm", 0.0);
        if (h->w_post) {
            RC(avx::layernorm(fast ? nullptr : w.f0, fast ? w.h0 : nullptr, D, h->ln0_w, h->ln0_b, 1e-5f, M, D, nullptr, D, w.h0, D, dt, cs));
        } else {   // embed_dim == encoder_embed_dim: the LayerNorm output is x itself; padded tokens are zeroed here (backbone.py:169-170)
            RC(avx::layernorm(fast ? nullptr : w.f0, fast ? w.h0 : nullptr, D, h->ln0_w, h->ln0_b, 1e-5f, M, D, fast ? nullptr : x32, E, w.xh, E, dt, cs));
            RC(avx::zero_rows(fast ? nullptr : x32, E, w.xh, E, M, E, pad, cs));
        }
        prof.end();
        if (h->w_post) {
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.h0; g.lda = D; g.W = h->w_post; g.ldw = D; g.M = M; g.N = E; g.K = D; g.bias = h->b_post;
            g.out_half = w.xh; g.ldh = E; g.row_zero = pad;
            if (!fast || hook0) { g.out_f32 = x32; g.ldo = E; }
            prof.begin("gemm.post_extract_proj", 2.0 * Md * E * D);
            RC(avx::gemm(g, dt, cs));
            prof.end();
        }
        if (hook0 && h->w_post) {
            // the reference's hook holds the tensor that the encoder then zeroes in place at padded tokens
            // (beats.py:359-361 + backbone.py:169-170), so the tap equals x after masking
            if (hook_pooled) RC(avx::mean_pool(x32, Bc, Tt, E, nullptr, hook_out[0] + (size_t)c0 * E, cs));
            else AVX_HIP_CHECK(hipMemcpyAsync(hook_out[0] + (size_t)c0 * Tt * E, x32, sizeof(float) * (size_t)M * E, hipMemcpyDeviceToDevice, cs));
        }
        // 3. convolutional positional embedding + residual, encoder LayerNorm
        prof.begin("posconv", 2.0 * Md * E * (E / h->cfg.conv_pos_groups) * h->cfg.conv_pos);
        RC(avx::posconv(w.xh, fast ? nullptr : x32, h->w_pc, h->b_pc, Bc, Tt, E, h->cfg.conv_pos_groups, h->cfg.conv_pos, pre32, preh, dt, cs));
        prof.end();
        prof.begin("layernorm", 0.0);
        RC(avx::layernorm(pre32, preh, E, h->lnE_w, h->lnE_b, 1e-5f, M, E, fast ? nullptr : x32, E, w.xh, E, dt, cs));
        prof.end();

        // 4. transformer layers (post-LN DeepNorm branch, backbone.py:350-375)
        // "fold": the two LayerNorms of a layer never run as kernels.  y1 = x*alpha + attn (preh) and y2 = x1*alpha + ffn (xh) stay raw
        // in the operand type with per-row partial statistics from the epilogue that wrote them; fc1 / the next QKV read them
        // through LayerNorm-folded weights, out_proj / fc2 apply LayerNorm to their residual on the fly (GemmArgs, gemm.hip).
        const bool fold = fast && h->ln_fold;      // any M: the same arithmetic whatever the chunking
        const int nseg = E / 64;
        for (int i = 0; i < L; ++i) {
            const Layer& ly = h->layers[i];
            const bool raw_in = fold && i > 0;      // xh holds y2 of layer i-1 (raw) instead of its LayerNorm
            const Layer* pl = i > 0 ? &h->layers[i - 1] : nullptr;
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.xh; g.lda = E; g.W = ly.w_qkv; g.ldw = E; g.M = M; g.N = 3 * E; g.K = E; g.bias = ly.b_qkv;
            g.out_half = w.qkv; g.ldh = 3 * E;
            if (raw_in) { g.W = ly.w_qkv_f; g.bias = ly.b_qkv_f; g.ln_rows = w.r2; g.ln_s = ly.s_qkv; }
            prof.begin("gemm.qkv", 2.0 * Md * 3 * E * E);
            RC(avx::gemm(g, dt, cs));
            prof.end();
            prof.begin("attention", 4.0 * Md * Tt * E + 2.0 * Md * 8 * (E / H) * H);
            RC(avx::attention(w.qkv, Bc, Tt, H, bias_tab, ly.grep_w, ly.grep_b, ly.grep_a, pad, w.ah, dt, cs));
            prof.end();
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.ah; g.lda = E; g.W = ly.w_o; g.ldw = E; g.M = M; g.N = E; g.K = E; g.bias = ly.b_o; g.alpha = h->alpha;
            if (fast) { g.resid_half = w.xh; g.ldrh = E; g.out_half = preh; g.ldh = E; }
            else { g.resid = x32; g.ldr = E; g.out_f32 = pre32; g.ldo = E; }
            if (fold) {
                g.stats_out = w.st1;
                if (raw_in) {
                    g.resid_half = nullptr; g.ldrh = 0;
                    g.lnr_y = w.xh; g.ldy = E; g.lnr_rows = w.r2; g.lnr_gamma = ly.ga_o; g.lnr_beta = ly.bb_o; g.lnr_prefolded = 1;
                }
            }
            prof.begin("gemm.out_proj", 2.0 * Md * E * E);
            RC(avx::gemm(g, dt, cs));
            prof.end();
            if (fold) {
                prof.begin("ln_rowstats", 0.0);
                RC(avx::ln_rowstats(w.st1, M, nseg, 1e-5f, w.r1, cs));
                prof.end();
            }
            if (!fold) {
                prof.begin("layernorm", 0.0);
                RC(avx::layernorm(pre32, preh, E, ly.ln1_w, ly.ln1_b, 1e-5f, M, E, fast ? nullptr : x32, E, w.xh, E, dt, cs));
                prof.end();
            }
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.xh; g.lda = E; g.W = ly.w_fc1; g.ldw = E; g.M = M; g.N = F; g.K = E; g.bias = ly.b_fc1; g.gelu = 1;
            g.out_half = w.hh; g.ldh = F;
            if (fold) { g.A = preh; g.W = ly.w_fc1_f; g.bias = ly.b_fc1_f; g.ln_rows = w.r1; g.ln_s = ly.s_fc1; }
            prof.begin("gemm.fc1", 2.0 * Md * F * E);
            RC(avx::gemm(g, dt, cs));
            prof.end();
            const bool hooked = (hook_mask >> (i + 1)) & 1u;
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.hh; g.lda = F; g.W = ly.w_fc2; g.ldw = F; g.M = M; g.N = E; g.K = F; g.bias = ly.b_fc2; g.alpha = h->alpha;
            if (fast) { g.resid_half = w.xh; g.ldrh = E; g.out_half = preh; g.ldh = E; }
            else { g.resid = x32; g.ldr = E; g.out_f32 = pre32; g.ldo = E; }
            if (fold) {   // residual = LN1(y1) on the fly; y2 (raw) goes to xh, which nothing reads any more in this layer
                g.resid_half = nullptr; g.ldrh = 0;
                g.lnr_y = preh; g.ldy = E; g.lnr_rows = w.r1; g.lnr_gamma = ly.ga_fc2; g.lnr_beta = ly.bb_fc2; g.lnr_prefolded = 1;
                g.out_half = w.xh; g.stats_out = (i + 1 < L) ? w.st2 : nullptr;      // the last layer's y2 goes to a LayerNorm kernel that takes its own statistics
            }
            if (hooked) {
                g.out_raw = hook_pooled ? w.raw : hook_out[i + 1] + (size_t)c0 * Tt * E;
                g.ldraw = E;
            }
            prof.begin("gemm.fc2", 2.0 * Md * E * F);
            RC(avx::gemm(g, dt, cs));
            prof.end();
            if (hooked && hook_pooled) RC(avx::mean_pool(w.raw, Bc, Tt, E, nullptr, hook_out[i + 1] + (size_t)c0 * E, cs));
            const bool last = i == L - 1;
            if (fold && !last) {
                prof.begin("ln_rowstats", 0.0);
                RC(avx::ln_rowstats(w.st2, M, nseg, 1e-5f, w.r2, cs));
                prof.end();
            }
            // the last LayerNorm produces the fp32 features (caller's buffer, or scratch when only pooling)
            float* xo = nullptr;
            if (last) xo = features_out ? features_out + (size_t)c0 * Tt * E : ((pooled_out || !fast) ? x32 : nullptr);
            else if (!fast) xo = x32;
            // pooled embedding only (the headline path): final LayerNorm and the mean over tokens in one pass, no fp32 feature tensor
            const bool fused_pool = last && pooled_out && !features_out && preh && !pre32 && E % 8 == 0 && E <= 768 && Bc >= 32;
            if (fused_pool) {      // the pre-LayerNorm sums y2 sit in preh, with the fold in xh
                prof.begin("layernorm+mean_pool", 0.0);
                RC(avx::layernorm_pool(fold ? w.xh : preh, E, ly.ln2_w, ly.ln2_b, 1e-5f, Bc, Tt, E, pooled_out + (size_t)c0 * E, dt, cs));
                prof.end();
            } else if (!fold) {
                prof.begin("layernorm", 0.0);
                if (xo || !last) RC(avx::layernorm(pre32, preh, E, ly.ln2_w, ly.ln2_b, 1e-5f, M, E, xo, E, last ? nullptr : w.xh, E, dt, cs));
                prof.end();
            } else if (last && xo) {   // the only LayerNorm of the layer stack that still runs: fp32 features from the raw y2
                prof.begin("layernorm", 0.0);
                RC(avx::layernorm(nullptr, w.xh, E, ly.ln2_w, ly.ln2_b, 1e-5f, M, E, xo, E, nullptr, E, dt, cs));
                prof.end();
            }
            if (last && pooled_out && !fused_pool) {
                prof.begin("mean_pool", 0.0);
                RC(avx::mean_pool(xo, Bc, Tt, E, nullptr, pooled_out + (size_t)c0 * E, cs));
                prof.end();
            }
        }
        if (L == 0) {
            // no layers: features = encoder LayerNorm output; recompute it in fp32 for the outputs
            if (features_out || pooled_out) {
                float* xo = features_out ? features_out + (size_t)c0 * Tt * E : x32;
                RC(avx::layernorm(pre32, preh, E, h->lnE_w, h->lnE_b, 1e-5f, M, E, xo, E, nullptr, E, dt, cs));
                if (pooled_out) RC(avx::mean_pool(xo, Bc, Tt, E, nullptr, pooled_out + (size_t)c0 * E, cs));
            }
        }
    }
    if (lanes > 1) {
        for (int i = 0; i + 1 < lanes; ++i) {
            if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
            AVX_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            AVX_HIP_CHECK(hipEventRecord(h->ev_join[i], h->side[i]));
            AVX_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[i], 0));
        }
    }
    }   // rounds
#undef RC
    if (h->d_ovf && h->h_ovf) AVX_HIP_CHECK(hipMemcpyAsync(h->h_ovf, h->d_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    if (h->profiling) {
        AVX_HIP_CHECK(hipStreamSynchronize(s));
        std::map<std::string, std::pair<double, double>> agg;  // name -> (ms, flops)
        std::vector<std::string> order;
        for (size_t i = 0; i < prof.next; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, h->recs[i].e0, h->recs[i].e1);
            if (!agg.count(h->recs[i].name)) order.push_back(h->recs[i].name);
            agg[h->recs[i].name].first += ms;
            agg[h->recs[i].name].second += h->recs[i].flops;
        }
        h->prof_names = order;
        h->prof_ms.clear(); h->prof_flops.clear(); h->prof_name_ptrs.clear();
        for (auto& nm : h->prof_names) {
            h->prof_ms.push_back((float)agg[nm].first);
            h->prof_flops.push_back(agg[nm].second);
        }
        for (auto& nm : h->prof_names) h->prof_name_ptrs.push_back(nm.c_str());
    }
    return AVEXHIP_OK;
}

}  // namespace

extern "C" avexhip_beats* avexhip_beats_create(const avexhip_beats_config* cfg, const avexhip_tensor* tensors, int n_tensors) {
    if (!cfg || !tensors || n_tensors <= 0) {
        avexhip_set_error("beats_create: null config or empty weight table");
        return nullptr;
    }
    if (avexhip_device_count() <= 0) {
        avexhip_set_error("beats_create: no HIP device visible (this path has no CPU fallback)");
        return nullptr;
    }
    const avexhip_beats_config& c = *cfg;
    if (c.encoder_attention_heads <= 0 || c.encoder_embed_dim != 64 * c.encoder_attention_heads) {
        avexhip_set_error("beats_create: head_dim must be 64 (E=%d, H=%d)", c.encoder_embed_dim, c.encoder_attention_heads);
        return nullptr;
    }
    if (c.encoder_embed_dim % 128 || c.encoder_ffn_embed_dim % 128 || c.embed_dim % 128 ||
        (c.input_patch_size * c.input_patch_size) % 64 || c.num_mel_bins % c.input_patch_size) {
        avexhip_set_error("beats_create: dims must be MFMA-tile multiples (E=%d F=%d D=%d P=%d mel=%d)", c.encoder_embed_dim,
                          c.encoder_ffn_embed_dim, c.embed_dim, c.input_patch_size, c.num_mel_bins);
        return nullptr;
    }
    if (c.conv_pos != 128 || c.encoder_embed_dim / (c.conv_pos_groups > 0 ? c.conv_pos_groups : 1) != 48) {
        avexhip_set_error("beats_create: positional conv must be k=128 with 48 channels/group (k=%d groups=%d)", c.conv_pos, c.conv_pos_groups);
        return nullptr;
    }
    if (c.encoder_layers < 0 || c.encoder_layers > 31) {
        avexhip_set_error("beats_create: encoder_layers=%d out of range", c.encoder_layers);
        return nullptr;
    }
    if (c.operand_dtype != AVEXHIP_F16 && c.operand_dtype != AVEXHIP_BF16) {
        avexhip_set_error("beats_create: unknown operand dtype %d", c.operand_dtype);
        return nullptr;
    }
    avexhip_beats* h = new avexhip_beats();
    h->cfg = c;
    h->dtype = c.operand_dtype;
    h->E = c.encoder_embed_dim; h->F = c.encoder_ffn_embed_dim; h->H = c.encoder_attention_heads;
    h->L = c.encoder_layers; h->D = c.embed_dim; h->P = c.input_patch_size; h->NM = c.num_mel_bins;
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 256;
    h->fast = c.residual_dtype != 0;
    {
        const char* e = getenv("AVEX_AMD_LN_FOLD");
        // The encoder's LayerNorms are folded into the GEMM epilogues around them (GemmArgs) unless AVEX_AMD_LN_FOLD=0: 24 LayerNorm
        // launches and 9 GB of traffic per 256-clip step disappear, +2.7 % (9 367 -> 9 623 clips/s alternating inside one process,
        // profiles/r03a_ln_fold.txt) and one rounding of the residual stream less per sublayer.
        h->ln_fold = h->fast && c.encoder_embed_dim % 256 == 0 && c.encoder_ffn_embed_dim % 256 == 0 && !(e && atoi(e) == 0);
    }
    {
        // Independent chunks of a batch can overlap on several HIP streams: one chunk's HBM-bound kernels
        // (LayerNorm, epilogue tails, attention staging) fill the gaps of another chunk's MFMA kernels.
        const char* e = getenv("AVEX_AMD_STREAMS");
        int ns = e ? atoi(e) : 1;
        h->nstreams = ns < 1 ? 1 : (ns > 4 ? 4 : ns);
        const char* fe = getenv("AVEX_AMD_FRONTEND_IN_LANE");
        h->fe_in_lane = !(fe && atoi(fe) == 0);
        for (int i = 0; i + 1 < h->nstreams; ++i) {
            if (hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess) {
                avexhip_set_error("beats_create: cannot create side streams");
                delete h;
                return nullptr;
            }
        }
        if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) {
            avexhip_set_error("beats_create: cannot create events");
            delete h;
            return nullptr;
        }
    }
    if (hipMalloc((void**)&h->d_ovf, sizeof(unsigned int)) != hipSuccess || hipMemset(h->d_ovf, 0, sizeof(unsigned int)) != hipSuccess ||
        hipHostMalloc((void**)&h->h_ovf, sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) {
        avexhip_set_error("beats_create: cannot allocate the range-alarm counter");
        delete h;
        return nullptr;
    }
    *h->h_ovf = 0;
    h->alpha = c.deep_norm ? powf(2.0f * (float)c.encoder_layers, 0.25f) : 1.0f;
    if (build(h, tensors, n_tensors) != AVEXHIP_OK) {
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void avexhip_beats_destroy(avexhip_beats* h) { delete h; }

extern "C" int avexhip_beats_num_tokens(const avexhip_beats* h, int64_t T) {
    if (!h) return 0;
    const int frames = avexhip_fbank_num_frames(h->fb, T);
    return (frames / h->P) * (h->NM / h->P);
}

extern "C" size_t avexhip_beats_workspace_bytes(const avexhip_beats* h, int B, int64_t T) {
    if (!h || B <= 0) return 0;
    const int Tt = avexhip_beats_num_tokens(h, T);
    if (Tt <= 0) return 0;
    int chunk = 1, lanes = 1;
    plan_chunks(h, B, Tt, &chunk, &lanes);
    return carve(h, nullptr, chunk, Tt).total * (size_t)lanes;
}

extern "C" int avexhip_beats_forward(avexhip_beats* h, const float* wav, int B, int64_t T, int64_t wav_stride,
                                     const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out, int hook_pooled,
                                     float* features_out, float* pooled_out, void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && wav, "beats_forward: null handle or input");
    AVX_REQUIRE(B > 0 && T > 0, "beats_forward: empty input B=%d T=%lld", B, (long long)T);
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_fbank_num_frames(h->fb, T);
    return forward_impl(h, wav, nullptr, B, T, wav_stride, frames, frame_pad, hook_mask, hook_out, hook_pooled, features_out,
                        pooled_out, workspace, ws_bytes, (hipStream_t)stream);
}

extern "C" int avexhip_beats_forward_fbank(avexhip_beats* h, const float* fbank, int B, int frames, const uint8_t* frame_pad,
                                           uint32_t hook_mask, float* const* hook_out, int hook_pooled, float* features_out,
                                           float* pooled_out, void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && fbank, "beats_forward_fbank: null handle or input");
    AVX_REQUIRE(B > 0 && frames > 0, "beats_forward_fbank: empty input");
    return forward_impl(h, nullptr, fbank, B, 0, 0, frames, frame_pad, hook_mask, hook_out, hook_pooled, features_out, pooled_out,
                        workspace, ws_bytes, (hipStream_t)stream);
}

extern "C" int avexhip_beats_overflow_count(avexhip_beats* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "overflow_count: null argument");
    if (synchronize) AVX_HIP_CHECK(hipStreamSynchronize((hipStream_t)sync_stream));
    *events = h->h_ovf ? *(volatile unsigned int*)h->h_ovf : 0u;
    return AVEXHIP_OK;
}

extern "C" int avexhip_beats_overflow_reset(avexhip_beats* h, void* stream) {
    AVX_REQUIRE(h, "overflow_reset: null handle");
    AVX_HIP_CHECK(hipMemsetAsync(h->d_ovf, 0, sizeof(unsigned int), (hipStream_t)stream));
    AVX_HIP_CHECK(hipMemcpyAsync(h->h_ovf, h->d_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AVEXHIP_OK;
}

extern "C" int avexhip_beats_set_profiling(avexhip_beats* h, int enabled) {
    AVX_REQUIRE(h, "set_profiling: null handle");
    h->profiling = enabled != 0;
    return AVEXHIP_OK;
}

extern "C" int avexhip_beats_last_profile(const avexhip_beats* h, const char* const** names, const float** ms,
                                          const double** flops, int* count) {
    AVX_REQUIRE(h && names && ms && flops && count, "last_profile: null argument");
    *names = h->prof_name_ptrs.data();
    *ms = h->prof_ms.data();
    *flops = h->prof_flops.data();
    *count = (int)h->prof_name_ptrs.size();
    return AVEXHIP_OK;
}
