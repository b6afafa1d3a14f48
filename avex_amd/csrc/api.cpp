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
#include "handle_core.h"

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
    g.pool_part = a->pool_part; g.pool_T = a->pool_rows; g.pool_mode = a->pool_mode;
    g.splitk_ws = a->splitk_ws; g.splitk_bytes = a->splitk_bytes;
    g.rows_out = a->rows_out; g.rows_eps = a->rows_eps;
    return avx::gemm(g, dtype, (hipStream_t)stream);
}
extern "C" int avexhip_pool_reduce(const float* part, int B, int T, int N, float* out, int64_t ldo, void* stream) {
    return avx::pool_reduce(part, B, T, N, out, ldo, (hipStream_t)stream);
}
extern "C" int avexhip_pool_reduce_mode(const float* part, int B, int T, int N, float* out, int64_t ldo, int mode, void* stream) {
    return avx::pool_reduce(part, B, T, N, out, ldo, (hipStream_t)stream, mode);
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
// BEATs handle (the shared parts -- weight table, layer parameters, the layer loop -- are in handle_core.h)
// ---------------------------------------------------------------------------------------------
const avx::FbankDev* avexhip_fbank_plan_dev(const avexhip_fbank_plan* plan);

using avxh::align_up;
using avxh::CoreCfg;
using avxh::CoreIo;
using avxh::CoreWs;
using avxh::dev_f32;
using avxh::dev_half;
using avxh::Layer;
using avxh::Prof;
using avxh::Table;

namespace {

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

struct avexhip_beats : avxh::HandleBase {
    avexhip_beats_config cfg;
    int E = 0, F = 0, H = 0, L = 0, D = 0, P = 0, NM = 0, chunk = 256;
    bool fast = false;   // residual stream / pre-LN sums in the operand type
    bool batch_invariant = false;  // residual_dtype bit 1: see CoreCfg::batch_invariant
    bool ln_fold = false;  // fast mode: LayerNorms between the GEMMs folded into their epilogues
    int ln_fold_min_rows = 0;   // ... for chunks of at least this many rows (avxh::fold_policy)
    int nstreams = 1;    // chunks of one forward run concurrently on this many streams (caller's + side streams)
    bool capturing = false;    // a forward being recorded into a hipGraph: one lane, no lazily created objects
    bool fe_in_lane = true;    // frontend of a chunk on the chunk's own lane stream (AVEX_AMD_FRONTEND_IN_LANE=0: all frontends before the fork)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    float alpha = 1.f;
    avexhip_fbank_plan* fb = nullptr;
    void* w_patch = nullptr; float* b_patch = nullptr;      // b_patch: conv_bias=True (beats.py:263-269), else NULL
    bool pre_ln = false; int act = 1; bool glu = false;     // layer_norm_first / activation_fn of the config (CoreCfg)
    int hidden_shift = 0;                                   // avexhip_beats_config::hidden_shift (CoreCfg)
    float* ln0_w = nullptr; float* ln0_b = nullptr;
    void* w_post = nullptr; float* b_post = nullptr;
    void* w_pc = nullptr; float* b_pc = nullptr;
    float* lnE_w = nullptr; float* lnE_b = nullptr;
    std::vector<Layer> layers;
    std::vector<float> rel_table;  // host [num_buckets, H]; empty if no relative position embedding
    // Toeplitz bias tables [H, 2T - 1], one per token count T, built ON THE FORWARD'S STREAM by a kernel from the resident bucket table
    // (no host build, no hipMalloc / hipMemcpy / hipFree inside a forward: a new clip length costs one 3 us kernel).  They live in an arena
    // allocated when the handle is created and are NEVER evicted or freed before the handle: a recorded hipGraph may hold a table's
    // address for as long as the handle lives (round 3's LRU eviction could free a table under a graph).  When the arena is full, a
    // forward builds its table in the caller's workspace instead (every time: nothing is remembered), so there is no limit on the
    // number of distinct lengths either.
    struct BiasTab { float* ptr; hipEvent_t ready; hipStream_t stream; bool done; };
    float* d_rel_table = nullptr;          // device copy of rel_table
    int* d_bucket_lut = nullptr;           // T5 bucket of every offset -lut_maxd .. lut_maxd (beyond: saturated), made on the host
    int lut_maxd = 0;
    char* bias_arena = nullptr; size_t bias_arena_bytes = 0, bias_arena_used = 0;
    std::map<int, BiasTab> bias_tabs;
    std::mutex bias_mu;

    CoreCfg core() const {
        CoreCfg c;
        c.E = E; c.F = F; c.H = H; c.L = L; c.alpha = alpha; c.eps = 1e-5f; c.hook_site = 0; c.fast = fast; c.fold = ln_fold;
        c.fold_min_rows = ln_fold_min_rows; c.batch_invariant = batch_invariant; c.act = act; c.glu = glu; c.hidden_shift = hidden_shift; c.pre_ln = pre_ln; c.final_ln_w = lnE_w; c.final_ln_b = lnE_b;
        return c;
    }
    ~avexhip_beats() override {
        for (auto& kv : bias_tabs) if (kv.second.ready) (void)hipEventDestroy(kv.second.ready);
        if (bias_arena) (void)hipFree(bias_arena);
        if (d_rel_table) (void)hipFree(d_rel_table);
        if (d_bucket_lut) (void)hipFree(d_bucket_lut);
        if (fb) avexhip_fbank_plan_destroy(fb);
        for (int i = 0; i < 3; ++i) { if (side[i]) (void)hipStreamDestroy(side[i]); if (ev_join[i]) (void)hipEventDestroy(ev_join[i]); }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

namespace {

// parameter names of avex/models/beats/backbone.py's encoder layers (state-dict keys under "backbone.")
const avxh::LayerNames BEATS_NAMES = {nullptr, "encoder.layers.%d.self_attn.q_proj", "encoder.layers.%d.self_attn.k_proj", "encoder.layers.%d.self_attn.v_proj",
                                      "encoder.layers.%d.self_attn.out_proj", "encoder.layers.%d.self_attn_layer_norm", "encoder.layers.%d.fc1",
                                      "encoder.layers.%d.fc2", "encoder.layers.%d.final_layer_norm", "encoder.layers.%d.self_attn.grep_linear",
                                      "encoder.layers.%d.self_attn.grep_a"};

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
    if (c.conv_bias) RC(dev_f32(h, tb, "patch_embedding.bias", D, &h->b_patch));
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
    {
        avxh::LayerNames nm = BEATS_NAMES;
        if (!c.gru_rel_pos) { nm.grep_linear = nullptr; nm.grep_a = nullptr; }
        const CoreCfg cc = h->core();
        for (int i = 0; i < h->L; ++i) RC(avxh::build_layer(h, tb, nm, cc, h->layers, i));
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
        // what the forwards build their bias tables from, resident: the bucket table and the bucket of every offset up to saturation
        // (the buckets come from avexhip_rel_bucket on the host -- the function the reference's golden buckets pin -- never from a
        // device logarithm)
        AVX_HIP_CHECK(hipMalloc((void**)&h->d_rel_table, sizeof(float) * h->rel_table.size()));
        AVX_HIP_CHECK(hipMemcpy(h->d_rel_table, h->rel_table.data(), sizeof(float) * h->rel_table.size(), hipMemcpyHostToDevice));
        const int last = c.num_buckets / 2 - 1;
        int maxd = c.max_distance > 1 ? c.max_distance : 1;
        while (maxd < (1 << 24) && !(avexhip_rel_bucket(-maxd, c.num_buckets, c.max_distance) == last &&
                                     avexhip_rel_bucket(maxd, c.num_buckets, c.max_distance) == c.num_buckets / 2 + last)) maxd *= 2;
        AVX_REQUIRE(maxd < (1 << 24), "beats_create: relative position buckets do not saturate (num_buckets=%d max_distance=%d)", c.num_buckets, c.max_distance);
        std::vector<int> lut((size_t)2 * maxd + 1);
        for (int d = -maxd; d <= maxd; ++d) lut[(size_t)(d + maxd)] = avexhip_rel_bucket(d, c.num_buckets, c.max_distance);
        h->lut_maxd = maxd;
        AVX_HIP_CHECK(hipMalloc((void**)&h->d_bucket_lut, sizeof(int) * lut.size()));
        AVX_HIP_CHECK(hipMemcpy(h->d_bucket_lut, lut.data(), sizeof(int) * lut.size(), hipMemcpyHostToDevice));
        const char* am = getenv("AVEX_AMD_BIAS_ARENA_MB");      // 0: no arena, every forward builds its table in the workspace
        long arena_mb = 16;
        if (am) {
            char* end = nullptr;
            arena_mb = strtol(am, &end, 10);
            if (end == am || *end != '\0' || arena_mb < 0 || arena_mb > 4096) {
                avexhip_set_error("beats_create: AVEX_AMD_BIAS_ARENA_MB must be an integer in 0..4096 (MiB), got '%s'", am);
                return AVEXHIP_ERR_INVALID;
            }
        }
        h->bias_arena_bytes = (size_t)arena_mb << 20;
        if (h->bias_arena_bytes) AVX_HIP_CHECK(hipMalloc((void**)&h->bias_arena, h->bias_arena_bytes));
    }
#undef RC
    AVX_HIP_CHECK(hipDeviceSynchronize());
    return AVEXHIP_OK;
}

// [H, 2T-1] Toeplitz rows of compute_bias (backbone.py:475-492) for a forward on stream `s`: the arena's table of this token count (built
// on `s` the first time it is asked for), or -- arena full -- a table built into `fallback` (the caller's workspace) for this forward only.
size_t bias_tab_bytes(const avexhip_beats* h, int T) { return h->rel_table.empty() ? 0 : sizeof(float) * (size_t)h->H * (size_t)(2 * T - 1); }

int bias_tab_for(avexhip_beats* h, int T, hipStream_t s, float* fallback, float** out) {
    *out = nullptr;
    if (h->rel_table.empty()) return AVEXHIP_OK;
    std::lock_guard<std::mutex> lk(h->bias_mu);
    auto capturing = [&]() {      // is `s` recording a graph right now?  (events of the arena's tables are not recorded into, or waited for inside, a capture)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
        return st == hipStreamCaptureStatusActive;
    };
    auto it = h->bias_tabs.find(T);
    bool use_arena = true;
    if (it != h->bias_tabs.end()) {
        avexhip_beats::BiasTab& e = it->second;
        // built on another stream and possibly still in flight: order this stream behind the build (once the build has completed, never again)
        if (!e.done && e.stream != s) {
            if (hipEventQuery(e.ready) == hipSuccess) e.done = true;
            else {
                (void)hipGetLastError();      // hipErrorNotReady is not an error
                if (capturing()) use_arena = false;
                else AVX_HIP_CHECK(hipStreamWaitEvent(s, e.ready, 0));
            }
        }
        if (use_arena) { *out = e.ptr; return AVEXHIP_OK; }
    }
    const size_t need = align_up(bias_tab_bytes(h, T));
    if (use_arena && h->bias_arena_used + need <= h->bias_arena_bytes && !capturing()) {
        avexhip_beats::BiasTab e{(float*)(h->bias_arena + h->bias_arena_used), nullptr, s, false};
        AVX_HIP_CHECK(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
        const int rc = avx::bias_toeplitz(h->d_rel_table, h->d_bucket_lut, h->lut_maxd, T, h->H, e.ptr, s);
        if (rc != AVEXHIP_OK) { (void)hipEventDestroy(e.ready); return rc; }
        AVX_HIP_CHECK(hipEventRecord(e.ready, s));
        h->bias_arena_used += need;
        h->bias_tabs[T] = e;
        *out = e.ptr;
        return AVEXHIP_OK;
    }
    AVX_REQUIRE(fallback, "beats_forward: no room for the relative position bias table (T=%d)", T);
    const int rc = avx::bias_toeplitz(h->d_rel_table, h->d_bucket_lut, h->lut_maxd, T, h->H, fallback, s);
    if (rc != AVEXHIP_OK) return rc;
    *out = fallback;
    return AVEXHIP_OK;
}

struct Ws {
    char* patches; float* f0; char* h0;
    CoreWs core;
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
    w.core = avxh::carve_core(h->core(), M, take);
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

int forward_impl(avexhip_beats* h, const float* wav, const float* fbank_in, int B, int64_t T, int64_t stride, int frames,
                 const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out, int hook_pooled,
                 float* features_out, float* pooled_out, void* workspace, size_t ws_bytes, hipStream_t s) {
    const int E = h->E, F = h->F, H = h->H, D = h->D, P = h->P, L = h->L, NM = h->NM, dt = h->dtype;
    const int nt = frames / P, nf = NM / P;
    const int Tt = nt * nf;
    AVX_REQUIRE(Tt >= 1, "beats_forward: input too short (%d frames -> 0 tokens)", frames);
    AVX_REQUIRE(hook_mask == 0 || hook_out, "beats_forward: hook_mask set but hook_out is NULL");
    AVX_REQUIRE(hook_pooled >= 0 && hook_pooled <= 3, "beats_forward: hook_pooled = %d (0 full taps, 1 mean, 2 max, 3 first token)", hook_pooled);
    AVX_REQUIRE((hook_mask >> (L + 1)) == 0, "beats_forward: hook_mask has bits beyond layer %d", L);
    // hook 0 is post_extract_proj's output; a model with embed_dim == encoder_embed_dim has no such layer (beats.py:357-358)
    AVX_REQUIRE(!(hook_mask & 1u) || h->w_post, "beats_forward: hook 0 (post_extract_proj) requested but this model has no post_extract_proj");
    for (int i = 0; i <= L; ++i)
        AVX_REQUIRE(!((hook_mask >> i) & 1u) || hook_out[i], "beats_forward: hook %d selected but hook_out[%d] is NULL", i, i);
    int chunk = 1, lanes = 1;
    plan_chunks(h, B, Tt, &chunk, &lanes);
    const Ws need = carve(h, nullptr, chunk, Tt);
    const size_t ws_need = need.total * (size_t)lanes + align_up(bias_tab_bytes(h, Tt));
    if (!workspace || ws_bytes < ws_need) {
        avexhip_set_error("beats_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, ws_need);
        return AVEXHIP_ERR_WORKSPACE;
    }
    float* bias_fallback = bias_tab_bytes(h, Tt) ? (float*)((char*)workspace + need.total * (size_t)lanes) : nullptr;
    if (h->profiling || h->capturing) lanes = 1;   // per-kernel event timing needs the kernels alone on the device; a captured forward is one stream
    float* bias_tab = nullptr;
    int rc = bias_tab_for(h, Tt, s, bias_fallback, &bias_tab);
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
        float* x32 = w.core.x;                       // fp32 x (precise mode; in fast mode only hook 0 / final output scratch)
        float* pre32 = fast ? nullptr : w.core.pre;
        void* preh = fast ? w.core.preh : nullptr;
        const bool hook0 = (hook_mask & 1u) != 0;
        avx::GemmArgs g;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.patches; g.lda = P * P; g.W = h->w_patch; g.ldw = P * P; g.M = M; g.N = D; g.K = P * P; g.bias = h->b_patch;
        if (fast) { g.out_half = w.h0; g.ldh = D; } else { g.out_f32 = w.f0; g.ldo = D; }
        prof.begin("gemm.patch_embed", 2.0 * Md * D * P * P);
        RC(avx::gemm(g, dt, cs));
        prof.end();
        prof.begin("layernorm", 0.0);
        if (h->w_post) {
            RC(avx::layernorm(fast ? nullptr : w.f0, fast ? w.h0 : nullptr, D, h->ln0_w, h->ln0_b, 1e-5f, M, D, nullptr, D, w.h0, D, dt, cs));
        } else {   // embed_dim == encoder_embed_dim: the LayerNorm output is x itself; padded tokens are zeroed here (backbone.py:169-170)
            RC(avx::layernorm(fast ? nullptr : w.f0, fast ? w.h0 : nullptr, D, h->ln0_w, h->ln0_b, 1e-5f, M, D, fast ? nullptr : x32, E, w.core.xh, E, dt, cs));
            RC(avx::zero_rows(fast ? nullptr : x32, E, w.core.xh, E, M, E, pad, cs));
        }
        prof.end();
        if (h->w_post) {
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = w.h0; g.lda = D; g.W = h->w_post; g.ldw = D; g.M = M; g.N = E; g.K = D; g.bias = h->b_post;
            g.out_half = w.core.xh; g.ldh = E; g.row_zero = pad;
            if (!fast || hook0) { g.out_f32 = x32; g.ldo = E; }
            prof.begin("gemm.post_extract_proj", 2.0 * Md * E * D);
            RC(avx::gemm(g, dt, cs));
            prof.end();
        }
        if (hook0 && h->w_post) {
            // the reference's hook holds the tensor that the encoder then zeroes in place at padded tokens
            // (beats.py:359-361 + backbone.py:169-170), so the tap equals x after masking
            if (hook_pooled) RC(avx::agg_pool(x32, Bc, Tt, E, hook_pooled, hook_out[0] + (size_t)c0 * E, cs));
            else AVX_HIP_CHECK(hipMemcpyAsync(hook_out[0] + (size_t)c0 * Tt * E, x32, sizeof(float) * (size_t)M * E, hipMemcpyDeviceToDevice, cs));
        }
        // 3. convolutional positional embedding + residual, encoder LayerNorm
        prof.begin("posconv", 2.0 * Md * E * (E / h->cfg.conv_pos_groups) * h->cfg.conv_pos);
        RC(avx::posconv(w.core.xh, fast ? nullptr : x32, h->w_pc, h->b_pc, Bc, Tt, E, h->cfg.conv_pos_groups, h->cfg.conv_pos, pre32, preh, dt, cs));
        prof.end();
        if (!h->pre_ln) {      // pre-LN models normalise after the stack instead (backbone.py:146-147, 176-177); their stream stays in pre32 / preh
            prof.begin("layernorm", 0.0);
            RC(avx::layernorm(pre32, preh, E, h->lnE_w, h->lnE_b, 1e-5f, M, E, fast ? nullptr : x32, E, w.core.xh, E, dt, cs));
            prof.end();
        }

        // 4. transformer layers (backbone.py:328-375, post-LN / DeepNorm or pre-LN): avxh::run_layers; hook i + 1 = layer i's raw fc2 output
        CoreIo io;
        io.Bc = Bc; io.Tt = Tt; io.c0 = (size_t)c0; io.bias_tab = bias_tab; io.pad = pad; io.hook_mask = hook_mask; io.hook_bit0 = 1;
        io.hook_out = hook_out; io.hook_pooled = hook_pooled; io.features_out = features_out; io.pooled_out = pooled_out;
        RC(avxh::run_layers(h, h->core(), h->layers, w.core, io, prof, cs));
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
    { const int rc2 = h->mirror_alarm(s); if (rc2 != AVEXHIP_OK) return rc2; }
    return prof.collect();
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
    if (c.activation_fn < AVEXHIP_FFN_GELU || c.activation_fn > AVEXHIP_FFN_GLU) {
        avexhip_set_error("beats_create: unknown activation_fn code %d", c.activation_fn);
        return nullptr;
    }
    if (c.layer_norm_first && c.deep_norm) {      // the reference asserts the same (beats.py:275)
        avexhip_set_error("beats_create: deep_norm and layer_norm_first exclude each other");
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
    if (c.hidden_shift < 0 || c.hidden_shift > 24 || (c.hidden_shift > 0 && c.activation_fn == AVEXHIP_FFN_GLU)) {
        avexhip_set_error("beats_create: hidden_shift must be 0..24 and is not built for the GLU feed-forward (got %d)", c.hidden_shift);
        return nullptr;
    }
    avexhip_beats* h = new avexhip_beats();
    h->cfg = c;
    h->dtype = c.operand_dtype;
    h->E = c.encoder_embed_dim; h->F = c.encoder_ffn_embed_dim; h->H = c.encoder_attention_heads;
    h->L = c.encoder_layers; h->D = c.embed_dim; h->P = c.input_patch_size; h->NM = c.num_mel_bins;
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 256;
    h->fast = avxh::cfg_fast(c.residual_dtype);
    h->batch_invariant = avxh::cfg_batch_invariant(c.residual_dtype);
    {
        // The encoder's LayerNorms are folded into the GEMM epilogues around them (GemmArgs) unless AVEX_AMD_LN_FOLD=0: 24 LayerNorm
        // launches and 9 GB of traffic per 256-clip step disappear, +2.7 % (9 367 -> 9 623 clips/s alternating inside one process,
        // profiles/r03a_ln_fold.txt) and one rounding of the residual stream less per sublayer.  Built for post-LN blocks with a GELU FFN.
        avxh::fold_policy(h->fast, c.encoder_embed_dim, c.encoder_ffn_embed_dim, &h->ln_fold, &h->ln_fold_min_rows, h->batch_invariant);
        if (c.layer_norm_first || c.activation_fn != AVEXHIP_FFN_GELU) h->ln_fold = false;
    }
    h->pre_ln = c.layer_norm_first != 0;
    h->glu = c.activation_fn == AVEXHIP_FFN_GLU;
    h->hidden_shift = c.hidden_shift;
    if (h->hidden_shift > 0) h->ln_fold = false;      // the folded epilogues are the fast ones, which do not scale (GemmArgs::half_scale)
    switch (c.activation_fn) {      // GemmArgs::gelu codes
        case AVEXHIP_FFN_GELU: h->act = 1; break;
        case AVEXHIP_FFN_RELU: h->act = 3; break;
        case AVEXHIP_FFN_GELU_TANH: h->act = 4; break;
        case AVEXHIP_FFN_TANH: h->act = 5; break;
        default: h->act = 0; break;      // linear; glu applies its own gate after fc1
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
    h->who = "beats_create";
    if (h->init_alarm() != AVEXHIP_OK) {
        delete h;
        return nullptr;
    }
    h->alpha = c.deep_norm ? powf(2.0f * (float)c.encoder_layers, 0.25f) : 1.0f;
    if (build(h, tensors, n_tensors) != AVEXHIP_OK || h->weights_fit() != AVEXHIP_OK) {
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
    return carve(h, nullptr, chunk, Tt).total * (size_t)lanes + align_up(bias_tab_bytes(h, Tt));      // + room for the bias table of a length the arena has no place for
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

// ---------------------------------------------------------------------------------------------
// A forward recorded as a hipGraph.  At small batch the path is launch-bound (about 95 kernels of a few microseconds each for one
// clip); replaying the recorded graph submits them in one call.
// ---------------------------------------------------------------------------------------------
struct avexhip_beats_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
    ~avexhip_beats_graph() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
};

extern "C" avexhip_beats_graph* avexhip_beats_graph_capture(avexhip_beats* h, const float* wav, int B, int64_t T, int64_t wav_stride,
                                                            const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out, int hook_pooled,
                                                            float* features_out, float* pooled_out, void* workspace, size_t ws_bytes, void* stream) {
    if (!h || !wav || B <= 0 || T <= 0) {
        avexhip_set_error("beats_graph_capture: null handle / input or empty batch");
        return nullptr;
    }
    if (h->profiling) {
        avexhip_set_error("beats_graph_capture: the handle is in profiling mode (events cannot be recorded into a graph)");
        return nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!s) {
        avexhip_set_error("beats_graph_capture: the default (NULL) stream cannot be captured; pass a stream created with hipStreamCreate");
        return nullptr;
    }
    if (wav_stride <= 0) wav_stride = T;
    const int frames = avexhip_fbank_num_frames(h->fb, T);
    // 1. an ordinary forward first: everything created lazily (the bias table of this token count, per-kernel LDS attributes, the
    //    compute-unit count) exists afterwards, so the recorded pass allocates nothing
    h->capturing = true;
    int rc = forward_impl(h, wav, nullptr, B, T, wav_stride, frames, frame_pad, hook_mask, hook_out, hook_pooled, features_out, pooled_out,
                          workspace, ws_bytes, s);
    if (rc == AVEXHIP_OK && hipStreamSynchronize(s) != hipSuccess) { avexhip_set_error("beats_graph_capture: the warm-up forward failed"); rc = AVEXHIP_ERR_HIP; }
    if (rc != AVEXHIP_OK) { h->capturing = false; return nullptr; }
    // 2. the same forward again, recorded
    avexhip_beats_graph* g = new avexhip_beats_graph();
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        avexhip_set_error("beats_graph_capture: hipStreamBeginCapture failed: %s", hipGetErrorString(e));
        h->capturing = false;
        delete g;
        return nullptr;
    }
    rc = forward_impl(h, wav, nullptr, B, T, wav_stride, frames, frame_pad, hook_mask, hook_out, hook_pooled, features_out, pooled_out,
                      workspace, ws_bytes, s);
    e = hipStreamEndCapture(s, &g->graph);      // always ends the capture, also after an error inside it
    h->capturing = false;
    if (rc != AVEXHIP_OK || e != hipSuccess || !g->graph) {
        if (rc == AVEXHIP_OK) avexhip_set_error("beats_graph_capture: hipStreamEndCapture failed: %s", hipGetErrorString(e));
        delete g;
        return nullptr;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        avexhip_set_error("beats_graph_capture: hipGraphInstantiate failed: %s", hipGetErrorString(e));
        delete g;
        return nullptr;
    }
    (void)hipGraphGetNodes(g->graph, nullptr, &g->nodes);
    return g;
}

extern "C" int avexhip_beats_graph_launch(avexhip_beats_graph* g, void* stream) {
    AVX_REQUIRE(g && g->exec, "beats_graph_launch: null graph");
    AVX_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return AVEXHIP_OK;
}

extern "C" int avexhip_beats_graph_nodes(const avexhip_beats_graph* g) { return g ? (int)g->nodes : 0; }

extern "C" void avexhip_beats_graph_destroy(avexhip_beats_graph* g) { delete g; }

extern "C" int avexhip_beats_overflow_count(avexhip_beats* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "overflow_count: null argument");
    return h->overflow_count(events, (hipStream_t)sync_stream, synchronize);
}

extern "C" int avexhip_beats_overflow_reset(avexhip_beats* h, void* stream) {
    AVX_REQUIRE(h, "overflow_reset: null handle");
    return h->overflow_reset((hipStream_t)stream);
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
