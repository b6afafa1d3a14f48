// Encoder handles of the other two transformer families on the path, with the BEATs handle's contract (caller-owned workspace,
// chunking, hook taps, range alarm, no hidden synchronisation):
//
//   avexhip_eat_*    EAT-base = the HF remote Data2Vec-multi image encoder the reference calls through
//                    EATHFModel.forward -> backbone.extract_features(spec[B, 1, 1024, 128]) (avex/models/eat_hf.py:201,241-289):
//                    filterbank -> 16 x 16 patch rows -> local_encoder GEMM -> class token + fixed positions + pre_norm ->
//                    12 post-LN blocks (hook = blocks.{i}.attn.proj's raw output) -> features [B, 513, 768], CLS / mean pooling.
//   avexhip_aves_*   AVES = torchaudio wav2vec2-base as aves_model.Model.forward calls it (avex/models/aves_model.py:62-151):
//                    7-layer convolutional feature extractor (layer 0: avexhip_wavconv0, layers 1-6: strided-row GEMMs) ->
//                    LayerNorm(512) -> Linear(512, 768) -> positional conv -> LayerNorm -> 12 post-LN layers (hook =
//                    feed_forward.output_dense's raw output) -> features [B, T', 768], mean pooling.
//
// Both run their layers through avxh::run_layers (handle_core.h) -- the BEATs loop with alpha = 1, no bias table and no gate -- so
// they get the folded LayerNorms and the streaming GEMM's epilogues too.  Round 2 composed these encoders in Python
// (avex_amd/eat_encoder.py, aves_encoder.py: one torch.empty per intermediate, no chunking); those classes are now thin wrappers.
// PARITY UNPINNED for both (third-party code absent from the reference tree): checkers oracle/eat_oracle.py, oracle/aves_oracle.py.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "handle_core.h"

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

int chunk_for(int max_chunk_clips, int B, int Tt) {
    // max_chunk_clips is sized for ~512-token clips; longer clips keep the same number of TOKEN rows per pass
    int cap = max_chunk_clips > 0 ? max_chunk_clips : 256;
    if (Tt > 520) { cap = (int)(((int64_t)cap * 512) / Tt); if (cap < 1) cap = 1; }
    return B < cap ? B : cap;
}


void hann_window(int win, std::vector<float>& w) {      // torch.hann_window(win, periodic=False), the kaldi "hanning" window
    w.resize(win);
    for (int n = 0; n < win; ++n) w[n] = 0.5f - 0.5f * cosf((float)n * (float)(M_PI * 2.0 / (double)(win - 1)));
}
void kaldi_mel(int n_fft, int n_mels, float sr, float low, float high, std::vector<float>& fb) {      // as api.cpp's default_mel (beats.py:82-118)
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
            fb[(size_t)k * n_mels + m] = fmaxf(0.f, fminf(up, down));
        }
    }
}

}  // namespace

// =============================================================================================
// EAT
// =============================================================================================
struct avexhip_eat : avxh::HandleBase {
    avexhip_eat_config cfg;
    CoreCfg core;
    int P = 16, n_patches = 0, chunk = 256;
    avexhip_fbank_plan* fb = nullptr;
    void* w_pe = nullptr; float* b_pe = nullptr;
    float* cls = nullptr; float* pos = nullptr;
    float* pre_w = nullptr; float* pre_b = nullptr;
    std::vector<Layer> layers;
    ~avexhip_eat() override { if (fb) avexhip_fbank_plan_destroy(fb); }
};

namespace {

const avxh::LayerNames EAT_NAMES = {"blocks.%d.attn.qkv", nullptr, nullptr, nullptr, "blocks.%d.attn.proj", "blocks.%d.norm1", "blocks.%d.mlp.fc1",
                                    "blocks.%d.mlp.fc2", "blocks.%d.norm2", nullptr, nullptr};

struct EatWs {
    char* patches; char* pe; float* clip_mean;
    CoreWs core;
    size_t total;
};

EatWs eat_carve(const avexhip_eat* h, char* base, int Bc) {
    const size_t Tt = (size_t)h->n_patches + 1;
    EatWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    w.patches = take((size_t)Bc * h->n_patches * h->P * h->P * 2);
    w.pe = take((size_t)Bc * h->n_patches * h->core.E * 2);
    w.clip_mean = (float*)take((size_t)Bc * 4);
    w.core = avxh::carve_core(h->core, (size_t)Bc * Tt, take);
    w.total = off;
    return w;
}

int eat_build(avexhip_eat* h, const avexhip_tensor* tensors, int n) {
    const avexhip_eat_config& c = h->cfg;
    Table tb{tensors, n};
    tb.strip2 = "model.";
    const int E = h->core.E;
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    {   // EATAudioProcessor's frontend (avex/models/eat/audio_processor.py:72-143): kaldi fbank, Hann window, no 2^15 scale, (x - mean) / (2 std)
        std::vector<float> hw, hm;
        hann_window(400, hw);
        kaldi_mel(512, c.n_mels, 16000.0f, 20.0f, 8000.0f, hm);
        avexhip_fbank_config fc;
        fc.win_length = 400; fc.hop_length = 160; fc.n_mels = c.n_mels;
        fc.input_scale = 1.0f; fc.preemph = 0.97f; fc.remove_dc = 1; fc.log_floor = 1.1920929e-07f;
        fc.norm_mean = c.norm_mean; fc.norm_div = 2.0f * c.norm_std;
        h->fb = avexhip_fbank_plan_create(&fc, hw.data(), hm.data());
        if (!h->fb) return AVEXHIP_ERR_HIP;
    }
    RC(dev_half(h, tb, "local_encoder.proj.weight", (int64_t)E * h->P * h->P, &h->w_pe));
    RC(dev_f32(h, tb, "local_encoder.proj.bias", E, &h->b_pe));
    RC(dev_f32(h, tb, "extra_tokens", E, &h->cls));
    {   // fixed 2-D sin/cos positions: the table may be longer than the image needs (one row per patch is used)
        const avexhip_tensor* t = tb.find("fixed_positional_encoder.positions");
        if (!t || !t->data || t->numel < (int64_t)h->n_patches * E || t->numel % E) {
            avexhip_set_error("eat_create: fixed_positional_encoder.positions missing or shorter than %d rows of %d", h->n_patches, E);
            return AVEXHIP_ERR_MISSING;
        }
        AVX_HIP_CHECK(hipMalloc((void**)&h->pos, sizeof(float) * (size_t)h->n_patches * E));
        h->allocs.push_back(h->pos);
        AVX_HIP_CHECK(hipMemcpy(h->pos, t->data, sizeof(float) * (size_t)h->n_patches * E, hipMemcpyDefault));
    }
    RC(dev_f32(h, tb, "pre_norm.weight", E, &h->pre_w));
    RC(dev_f32(h, tb, "pre_norm.bias", E, &h->pre_b));
    h->layers.resize(h->core.L);
    for (int i = 0; i < h->core.L; ++i) RC(avxh::build_layer(h, tb, EAT_NAMES, h->core, h->layers, i));
#undef RC
    AVX_HIP_CHECK(hipDeviceSynchronize());
    return AVEXHIP_OK;
}

}  // namespace

extern "C" avexhip_eat* avexhip_eat_create(const avexhip_eat_config* cfg, const avexhip_tensor* tensors, int n_tensors) {
    if (!cfg || !tensors || n_tensors <= 0) { avexhip_set_error("eat_create: null config or empty weight table"); return nullptr; }
    if (avexhip_device_count() <= 0) { avexhip_set_error("eat_create: no HIP device visible (this path has no CPU fallback)"); return nullptr; }
    const avexhip_eat_config& c = *cfg;
    if (c.num_heads <= 0 || c.embed_dim != 64 * c.num_heads) { avexhip_set_error("eat_create: head_dim must be 64 (E=%d, H=%d)", c.embed_dim, c.num_heads); return nullptr; }
    if (c.embed_dim % 128 || c.ffn_dim % 128) { avexhip_set_error("eat_create: dims must be MFMA-tile multiples (E=%d F=%d)", c.embed_dim, c.ffn_dim); return nullptr; }
    if (c.patch_size != 16 || c.n_mels % 16 || c.target_length % 16 || c.n_mels <= 0 || c.target_length <= 0) {
        avexhip_set_error("eat_create: only 16 x 16 patches over a (16 a) x (16 b) image are built (patch %d, image %d x %d)", c.patch_size, c.target_length, c.n_mels);
        return nullptr;
    }
    if (c.depth < 0 || c.depth > 32) { avexhip_set_error("eat_create: depth=%d out of range", c.depth); return nullptr; }
    if (c.operand_dtype != AVEXHIP_F16 && c.operand_dtype != AVEXHIP_BF16) { avexhip_set_error("eat_create: unknown operand dtype %d", c.operand_dtype); return nullptr; }
    if (c.norm_mean == 0.0f && c.norm_std == 1.0f) {
        avexhip_set_error("eat_create: per-sample normalisation (norm_mean 0, norm_std 1) is not built in the fused path");
        return nullptr;
    }
    avexhip_eat* h = new avexhip_eat();
    h->who = "eat_create";
    h->cfg = c;
    h->dtype = c.operand_dtype;
    h->core.E = c.embed_dim; h->core.F = c.ffn_dim; h->core.H = c.num_heads; h->core.L = c.depth;
    h->core.alpha = 1.0f; h->core.eps = c.norm_eps > 0.f ? c.norm_eps : 1e-6f; h->core.hook_site = 1;
    h->core.fast = avxh::cfg_fast(c.residual_dtype);
    h->core.batch_invariant = avxh::cfg_batch_invariant(c.residual_dtype);
    avxh::fold_policy(h->core.fast, c.embed_dim, c.ffn_dim, &h->core.fold, &h->core.fold_min_rows, h->core.batch_invariant);
    h->P = c.patch_size;
    h->n_patches = (c.target_length / 16) * (c.n_mels / 16);
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 256;
    if (h->init_alarm() != AVEXHIP_OK || eat_build(h, tensors, n_tensors) != AVEXHIP_OK || h->weights_fit() != AVEXHIP_OK) { delete h; return nullptr; }
    return h;
}

extern "C" void avexhip_eat_destroy(avexhip_eat* h) { delete h; }
extern "C" int avexhip_eat_num_tokens(const avexhip_eat* h) { return h ? h->n_patches + 1 : 0; }

extern "C" size_t avexhip_eat_workspace_bytes(const avexhip_eat* h, int B) {
    if (!h || B <= 0) return 0;
    return eat_carve(h, nullptr, chunk_for(h->chunk, B, h->n_patches + 1)).total;
}

extern "C" int avexhip_eat_forward(avexhip_eat* h, const float* wav, int B, int64_t T, int64_t wav_stride, const float* spec,
                                   uint32_t hook_mask, float* const* hook_out, int hook_pooled, float* features_out, float* pooled_out,
                                   int pooling, void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && ((wav != nullptr) != (spec != nullptr)), "eat_forward: give exactly one of wav / spec");
    AVX_REQUIRE(B > 0 && (spec || T > 0), "eat_forward: empty input B=%d T=%lld", B, (long long)T);
    AVX_REQUIRE(pooling >= 0 && pooling <= 2 && (pooling == 0) == (pooled_out == nullptr), "eat_forward: pooling (0 none, 1 cls, 2 mean) and pooled_out must agree");
    const int E = h->core.E, L = h->core.L, dt = h->dtype, Tp = h->n_patches, Tt = Tp + 1, PP = h->P * h->P;
    AVX_REQUIRE(hook_mask == 0 || hook_out, "eat_forward: hook_mask set but hook_out is NULL");
    AVX_REQUIRE(L >= 32 || (hook_mask >> L) == 0, "eat_forward: hook_mask has bits beyond block %d", L - 1);
    for (int i = 0; i < L; ++i) AVX_REQUIRE(!((hook_mask >> i) & 1u) || hook_out[i], "eat_forward: hook %d selected but hook_out[%d] is NULL", i, i);
    AVX_REQUIRE(L > 0, "eat_forward: a model without blocks has no output");
    hipStream_t s = (hipStream_t)stream;
    if (wav && wav_stride <= 0) wav_stride = T;
    const int chunk = chunk_for(h->chunk, B, Tt);
    const EatWs need = eat_carve(h, nullptr, chunk);
    if (!workspace || ws_bytes < need.total) {
        avexhip_set_error("eat_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, need.total);
        return AVEXHIP_ERR_WORKSPACE;
    }
    const avx::FbankDev* fbd = avexhip_fbank_plan_dev(h->fb);
    Prof prof{h, s};
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
        const EatWs w = eat_carve(h, (char*)workspace, chunk);
        const double Md = (double)Bc * Tt;
        // 1. the log-mel image as 16 x 16 patch rows, straight from the filterbank (clip mean removed first, audio_processor.py:107;
        //    rows past the last frame are the normalised zero padding, :121-135) -- or cut from a caller-made image
        if (wav) {
            const int frames = avexhip_fbank_num_frames(h->fb, T);
            prof.begin("fbank", (double)Bc * frames * (5.0 * 512 * 9 + 2.0 * 504));
            RC(avexhip_clip_mean(wav + (size_t)c0 * wav_stride, Bc, T, wav_stride, w.clip_mean, s));
            RC(avx::fbank(*fbd, wav + (size_t)c0 * wav_stride, Bc, T, wav_stride, frames, nullptr, w.patches, h->P, dt, s, w.clip_mean, h->cfg.target_length));
            prof.end();
        } else {
            prof.begin("patchify", 0.0);
            RC(avx::patchify(spec + (size_t)c0 * h->cfg.target_length * h->cfg.n_mels, Bc, h->cfg.target_length, h->cfg.n_mels, h->P, w.patches, dt, s));
            prof.end();
        }
        // 2. local_encoder (Conv2d(1, E, 16, 16) as a GEMM), then class token + fixed positions + pre_norm
        avx::GemmArgs g;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.patches; g.lda = PP; g.W = h->w_pe; g.ldw = PP; g.M = Bc * Tp; g.N = E; g.K = PP; g.bias = h->b_pe;
        g.out_half = w.pe; g.ldh = E;
        prof.begin("gemm.patch_embed", 2.0 * Bc * Tp * (double)E * PP);
        RC(avx::gemm(g, dt, s));
        prof.end();
        prof.begin("token_embed_ln", 0.0);
        RC(avx::token_embed_ln(w.pe, h->pos, h->cls, h->pre_w, h->pre_b, h->core.eps, Bc, Tp, E, w.core.xh, h->core.fast ? nullptr : w.core.x, dt, s));
        prof.end();
        // 3. the blocks; hook i = blocks.{i}.attn.proj (eat_hf.py:220-236)
        CoreIo io;
        io.Bc = Bc; io.Tt = Tt; io.c0 = (size_t)c0; io.hook_mask = hook_mask; io.hook_bit0 = 0; io.hook_out = hook_out; io.hook_pooled = hook_pooled < 0 ? 0 : (hook_pooled > 3 ? 3 : hook_pooled);
        io.features_out = features_out;
        io.pooled_out = pooling == 2 ? pooled_out : nullptr;
        float* cls_scratch = nullptr;
        if (pooling == 1 && !features_out) {      // CLS pooling without the feature tensor: the final LayerNorm goes to scratch, row 0 of every clip is kept
            cls_scratch = w.core.x;
            io.features_out = cls_scratch - (size_t)c0 * Tt * E;        // run_layers adds c0 * Tt * E
        }
        RC(avxh::run_layers(h, h->core, h->layers, w.core, io, prof, s));
        if (pooling == 1)
            AVX_HIP_CHECK(hipMemcpy2DAsync(pooled_out + (size_t)c0 * E, sizeof(float) * E, io.final_f32, sizeof(float) * (size_t)Tt * E, sizeof(float) * E, Bc,
                                           hipMemcpyDeviceToDevice, s));
        (void)Md;
    }
#undef RC
    { const int rc2 = h->mirror_alarm(s); if (rc2 != AVEXHIP_OK) return rc2; }
    return prof.collect();
}

extern "C" int avexhip_eat_overflow_count(avexhip_eat* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "eat_overflow_count: null argument");
    return h->overflow_count(events, (hipStream_t)sync_stream, synchronize);
}
extern "C" int avexhip_eat_set_profiling(avexhip_eat* h, int enabled) {
    AVX_REQUIRE(h, "eat_set_profiling: null handle");
    h->profiling = enabled != 0;
    return AVEXHIP_OK;
}
extern "C" int avexhip_eat_last_profile(const avexhip_eat* h, const char* const** names, const float** ms, const double** flops, int* count) {
    AVX_REQUIRE(h && names && ms && flops && count, "eat_last_profile: null argument");
    *names = h->prof_name_ptrs.data(); *ms = h->prof_ms.data(); *flops = h->prof_flops.data(); *count = (int)h->prof_name_ptrs.size();
    return AVEXHIP_OK;
}

// =============================================================================================
// A stack of post-LN transformer layers on caller-provided token rows (the TransformerProbe's nn.TransformerEncoder)
// =============================================================================================
struct avexhip_stack : avxh::HandleBase {
    avexhip_stack_config cfg;
    CoreCfg core;
    int chunk = 256;
    std::vector<Layer> layers;
};

namespace {
// torch.nn.TransformerEncoderLayer's parameter names, with in_proj_weight / in_proj_bias passed as in_proj.weight / in_proj.bias
const avxh::LayerNames STACK_NAMES = {"layers.%d.self_attn.in_proj", nullptr, nullptr, nullptr, "layers.%d.self_attn.out_proj", "layers.%d.norm1",
                                      "layers.%d.linear1", "layers.%d.linear2", "layers.%d.norm2", nullptr, nullptr};
struct StackWs { CoreWs core; size_t total; };
StackWs stack_carve(const avexhip_stack* h, char* base, size_t M) {
    StackWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    w.core = avxh::carve_core(h->core, M, take);
    w.total = off;
    return w;
}
}  // namespace

extern "C" avexhip_stack* avexhip_stack_create(const avexhip_stack_config* cfg, const avexhip_tensor* tensors, int n_tensors) {
    if (!cfg || !tensors || n_tensors <= 0) { avexhip_set_error("stack_create: null config or empty weight table"); return nullptr; }
    if (avexhip_device_count() <= 0) { avexhip_set_error("stack_create: no HIP device visible (this path has no CPU fallback)"); return nullptr; }
    const avexhip_stack_config& c = *cfg;
    const int hd = c.num_heads > 0 ? c.embed_dim / c.num_heads : 0;
    if (c.num_heads <= 0 || c.embed_dim != hd * c.num_heads || (hd != 32 && hd != 64 && hd != 96 && hd != 128)) {
        avexhip_set_error("stack_create: head width must be 32, 64, 96 or 128 (E=%d, H=%d)", c.embed_dim, c.num_heads);
        return nullptr;
    }
    if (c.embed_dim % 128 || c.ffn_dim % 128 || c.ffn_dim < 0) { avexhip_set_error("stack_create: dims must be MFMA-tile multiples (E=%d F=%d)", c.embed_dim, c.ffn_dim); return nullptr; }
    if (c.num_layers < 1 || c.num_layers > 32) { avexhip_set_error("stack_create: num_layers=%d out of range", c.num_layers); return nullptr; }
    if (c.activation != 1 && c.activation != 3) { avexhip_set_error("stack_create: activation %d (1 = erf GELU, 3 = ReLU)", c.activation); return nullptr; }
    if (c.operand_dtype != AVEXHIP_F16 && c.operand_dtype != AVEXHIP_BF16) { avexhip_set_error("stack_create: unknown operand dtype %d", c.operand_dtype); return nullptr; }
    avexhip_stack* h = new avexhip_stack();
    h->who = "stack_create";
    h->cfg = c;
    h->dtype = c.operand_dtype;
    h->core.E = c.embed_dim; h->core.F = c.ffn_dim; h->core.H = c.num_heads; h->core.L = c.num_layers; h->core.head_dim = hd;
    h->core.alpha = 1.0f; h->core.eps = c.norm_eps > 0.f ? c.norm_eps : 1e-5f; h->core.hook_site = 0;
    h->core.fast = avxh::cfg_fast(c.residual_dtype);
    h->core.batch_invariant = avxh::cfg_batch_invariant(c.residual_dtype);
    h->core.act = c.activation;
    avxh::fold_policy(h->core.fast, c.embed_dim, c.ffn_dim, &h->core.fold, &h->core.fold_min_rows, h->core.batch_invariant);
    if (c.activation != 1 || c.ffn_dim == 0) h->core.fold = false;      // the folded epilogues know GELU only
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 256;
    if (h->init_alarm() != AVEXHIP_OK) { delete h; return nullptr; }
    const Table tb{tensors, n_tensors};
    h->layers.resize(h->core.L);
    for (int i = 0; i < h->core.L; ++i)
        if (avxh::build_layer(h, tb, STACK_NAMES, h->core, h->layers, i) != AVEXHIP_OK) { delete h; return nullptr; }
    if (hipDeviceSynchronize() != hipSuccess) { avexhip_set_error("stack_create: upload failed"); delete h; return nullptr; }
    if (h->weights_fit() != AVEXHIP_OK) { delete h; return nullptr; }
    return h;
}

extern "C" void avexhip_stack_destroy(avexhip_stack* h) { delete h; }

extern "C" size_t avexhip_stack_workspace_bytes(const avexhip_stack* h, int B, int T) {
    if (!h || B <= 0 || T <= 0) return 0;
    return stack_carve(h, nullptr, (size_t)chunk_for(h->chunk, B, T) * T).total;
}

extern "C" int avexhip_stack_forward(avexhip_stack* h, const float* x, int B, int T, const uint8_t* key_pad, float* features_out, float* pooled_out,
                                     void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && x && B > 0 && T > 0, "stack_forward: null handle / input or empty batch");
    AVX_REQUIRE(features_out || pooled_out, "stack_forward: no output requested");
    const int E = h->core.E, dt = h->dtype;
    hipStream_t s = (hipStream_t)stream;
    const int chunk = chunk_for(h->chunk, B, T);
    const StackWs need = stack_carve(h, nullptr, (size_t)chunk * T);
    if (!workspace || ws_bytes < need.total) {
        avexhip_set_error("stack_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, need.total);
        return AVEXHIP_ERR_WORKSPACE;
    }
    Prof prof{h, s};
    int rc;
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
        const StackWs w = stack_carve(h, (char*)workspace, (size_t)chunk * T);
        const float* xc = x + (size_t)c0 * T * E;
        const size_t n = (size_t)Bc * T * E;
        prof.begin("cast", 0.0);
        rc = avx::cast_to_half(xc, w.core.xh, (int64_t)n, dt, s);
        if (rc != AVEXHIP_OK) return rc;
        if (!h->core.fast) AVX_HIP_CHECK(hipMemcpyAsync(w.core.x, xc, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
        prof.end();
        CoreIo io;
        io.Bc = Bc; io.Tt = T; io.c0 = (size_t)c0; io.pad = key_pad ? key_pad + (size_t)c0 * T : nullptr;
        io.features_out = features_out; io.pooled_out = pooled_out;
        rc = avxh::run_layers(h, h->core, h->layers, w.core, io, prof, s);
        if (rc != AVEXHIP_OK) return rc;
    }
    { const int rc2 = h->mirror_alarm(s); if (rc2 != AVEXHIP_OK) return rc2; }
    return prof.collect();
}

extern "C" int avexhip_stack_overflow_count(avexhip_stack* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "stack_overflow_count: null argument");
    return h->overflow_count(events, (hipStream_t)sync_stream, synchronize);
}

// =============================================================================================
// AVES (wav2vec2-base)
// =============================================================================================
struct avexhip_aves : avxh::HandleBase {
    avexhip_aves_config cfg;
    CoreCfg core;
    int chunk = 64, n_conv = 7;
    int ck[8], cs[8];                       // kernel / stride of every conv layer
    float* w0 = nullptr; float* gn_w = nullptr; float* gn_b = nullptr;
    void* wc[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float* zero_bias = nullptr;
    float* fp_ln_w = nullptr; float* fp_ln_b = nullptr;
    void* fp_w = nullptr; float* fp_b = nullptr;
    void* pc_w = nullptr; float* pc_b = nullptr;
    float* enc_ln_w = nullptr; float* enc_ln_b = nullptr;
    std::vector<Layer> layers;
};

namespace {

const avxh::LayerNames AVES_NAMES = {nullptr, "encoder.transformer.layers.%d.attention.q_proj", "encoder.transformer.layers.%d.attention.k_proj",
                                     "encoder.transformer.layers.%d.attention.v_proj", "encoder.transformer.layers.%d.attention.out_proj",
                                     "encoder.transformer.layers.%d.layer_norm", "encoder.transformer.layers.%d.feed_forward.intermediate_dense",
                                     "encoder.transformer.layers.%d.feed_forward.output_dense", "encoder.transformer.layers.%d.final_layer_norm", nullptr, nullptr};
constexpr int CC = 512;                     // channels of every conv layer (wav2vec2-base)
constexpr int SLACK = 8;                    // rows past a buffer's last clip that the next layer's strided rows may touch

// valid frames F_l and padded per-clip row counts P_l of every conv layer for T samples: P_{l-1} = stride_l * P_l >= F_{l-1}, so that
// one uniform lda covers the whole batch (avex_amd/aves_encoder.py conv_frame_plan)
int frame_plan(const avexhip_aves* h, int64_t T, int* F, int* P) {
    int64_t n = T;
    for (int l = 0; l < h->n_conv; ++l) {
        n = n >= h->ck[l] ? (n - h->ck[l]) / h->cs[l] + 1 : 0;
        F[l] = (int)n;
    }
    if (F[h->n_conv - 1] <= 0) return -1;
    for (int pad = 0;; ++pad) {
        P[h->n_conv - 1] = F[h->n_conv - 1] + pad;
        for (int l = h->n_conv - 1; l > 0; --l) P[l - 1] = h->cs[l] * P[l];
        bool ok = true;
        for (int l = 0; l < h->n_conv; ++l) ok = ok && P[l] >= F[l];
        if (ok) return 0;
    }
}

struct AvesWs {
    char* conv[2]; float* stats; char* feats; char* h0;
    CoreWs core;
    size_t total;
};

AvesWs aves_carve(const avexhip_aves* h, char* base, int Bc, int64_t T, const int* F, const int* P) {
    AvesWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    const size_t Tt = (size_t)F[h->n_conv - 1];
    w.conv[0] = take(((size_t)Bc * P[0] + SLACK) * CC * 2);
    w.conv[1] = take(((size_t)Bc * P[1] + SLACK) * CC * 2);
    w.stats = (float*)take((size_t)avexhip_wavconv0_stats_floats(Bc, T) * 4);
    w.feats = take((size_t)Bc * Tt * CC * 2);
    w.h0 = take((size_t)Bc * Tt * CC * 2);
    w.core = avxh::carve_core(h->core, (size_t)Bc * Tt, take);
    w.total = off;
    return w;
}

int aves_build(avexhip_aves* h, const avexhip_tensor* tensors, int n) {
    const avexhip_aves_config& c = h->cfg;
    Table tb{tensors, n};
    tb.strip1 = "model."; tb.strip2 = nullptr;
    const int E = h->core.E;
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    const std::string fe = "feature_extractor.conv_layers.";
    RC(dev_f32(h, tb, fe + "0.conv.weight", (int64_t)CC * h->ck[0], &h->w0));
    RC(dev_f32(h, tb, fe + "0.layer_norm.weight", CC, &h->gn_w));
    RC(dev_f32(h, tb, fe + "0.layer_norm.bias", CC, &h->gn_b));
    for (int l = 1; l < h->n_conv; ++l) {
        // conv weight [out, in, k] -> [out, k, in]: the K order of a strided activation row is (frame, channel)
        const int k = h->ck[l];
        std::vector<float> src, dst((size_t)CC * CC * k);
        RC(avxh::host_f32(h, tb, fe + std::to_string(l) + ".conv.weight", (int64_t)CC * CC * k, src));
        for (int o = 0; o < CC; ++o)
            for (int i = 0; i < CC; ++i)
                for (int t = 0; t < k; ++t) dst[((size_t)o * k + t) * CC + i] = src[((size_t)o * CC + i) * k + t];
        AVX_HIP_CHECK(hipMalloc(&h->wc[l], 2 * dst.size()));
        h->allocs.push_back(h->wc[l]);
        RC(avxh::upload_half(h, dst.data(), (int64_t)dst.size(), h->wc[l], "conv weight"));
    }
    AVX_HIP_CHECK(hipMalloc((void**)&h->zero_bias, sizeof(float) * CC));
    h->allocs.push_back(h->zero_bias);
    AVX_HIP_CHECK(hipMemset(h->zero_bias, 0, sizeof(float) * CC));
    RC(dev_f32(h, tb, "encoder.feature_projection.layer_norm.weight", CC, &h->fp_ln_w));
    RC(dev_f32(h, tb, "encoder.feature_projection.layer_norm.bias", CC, &h->fp_ln_b));
    RC(dev_half(h, tb, "encoder.feature_projection.projection.weight", (int64_t)E * CC, &h->fp_w));
    RC(dev_f32(h, tb, "encoder.feature_projection.projection.bias", E, &h->fp_b));
    {   // positional conv: fold weight-norm and repack (parametrizations.* of torch >= 2.1, weight_g / weight_v before)
        const int K = c.pos_conv_kernel, G = c.pos_conv_groups, cg = E / G;
        const std::string t = "encoder.transformer.pos_conv_embed.conv.";
        float *g = nullptr, *v = nullptr;
        const bool newer = tb.find(t + "parametrizations.weight.original0") != nullptr;
        RC(dev_f32(h, tb, t + (newer ? "parametrizations.weight.original0" : "weight_g"), K, &g));
        RC(dev_f32(h, tb, t + (newer ? "parametrizations.weight.original1" : "weight_v"), (int64_t)E * cg * K, &v));
        AVX_HIP_CHECK(hipMalloc(&h->pc_w, 2 * (size_t)E * cg * K));
        h->allocs.push_back(h->pc_w);
        RC(avx::posconv_pack(g, v, E, G, K, h->pc_w, h->dtype, nullptr));
        RC(dev_f32(h, tb, t + "bias", E, &h->pc_b));
    }
    RC(dev_f32(h, tb, "encoder.transformer.layer_norm.weight", E, &h->enc_ln_w));
    RC(dev_f32(h, tb, "encoder.transformer.layer_norm.bias", E, &h->enc_ln_b));
    h->layers.resize(h->core.L);
    for (int i = 0; i < h->core.L; ++i) RC(avxh::build_layer(h, tb, AVES_NAMES, h->core, h->layers, i));
#undef RC
    AVX_HIP_CHECK(hipDeviceSynchronize());
    return AVEXHIP_OK;
}

}  // namespace

extern "C" avexhip_aves* avexhip_aves_create(const avexhip_aves_config* cfg, const avexhip_tensor* tensors, int n_tensors) {
    if (!cfg || !tensors || n_tensors <= 0) { avexhip_set_error("aves_create: null config or empty weight table"); return nullptr; }
    if (avexhip_device_count() <= 0) { avexhip_set_error("aves_create: no HIP device visible (this path has no CPU fallback)"); return nullptr; }
    const avexhip_aves_config& c = *cfg;
    if (c.num_heads <= 0 || c.embed_dim != 64 * c.num_heads) { avexhip_set_error("aves_create: head_dim must be 64 (E=%d, H=%d)", c.embed_dim, c.num_heads); return nullptr; }
    if (c.embed_dim % 128 || c.ffn_dim % 128) { avexhip_set_error("aves_create: dims must be MFMA-tile multiples (E=%d F=%d)", c.embed_dim, c.ffn_dim); return nullptr; }
    if (c.n_conv_layers < 2 || c.n_conv_layers > 8 || c.conv_kernel[0] != 10 || c.conv_stride[0] != 5) {
        avexhip_set_error("aves_create: only the wav2vec2-base feature extractor layout (512 channels, first layer k=10 s=5) is built");
        return nullptr;
    }
    if (c.pos_conv_kernel != 128 || c.embed_dim / (c.pos_conv_groups > 0 ? c.pos_conv_groups : 1) != 48) {
        avexhip_set_error("aves_create: positional conv must be k=128 with 48 channels/group (k=%d groups=%d)", c.pos_conv_kernel, c.pos_conv_groups);
        return nullptr;
    }
    if (c.num_layers < 1 || c.num_layers > 32) { avexhip_set_error("aves_create: num_layers=%d out of range", c.num_layers); return nullptr; }
    if (c.operand_dtype != AVEXHIP_F16 && c.operand_dtype != AVEXHIP_BF16) { avexhip_set_error("aves_create: unknown operand dtype %d", c.operand_dtype); return nullptr; }
    avexhip_aves* h = new avexhip_aves();
    h->who = "aves_create";
    h->cfg = c;
    h->dtype = c.operand_dtype;
    h->n_conv = c.n_conv_layers;
    for (int l = 0; l < h->n_conv; ++l) {
        h->ck[l] = c.conv_kernel[l]; h->cs[l] = c.conv_stride[l];
        if (h->ck[l] <= 0 || h->cs[l] <= 0) { avexhip_set_error("aves_create: conv layer %d has kernel %d stride %d", l, h->ck[l], h->cs[l]); delete h; return nullptr; }
    }
    h->core.E = c.embed_dim; h->core.F = c.ffn_dim; h->core.H = c.num_heads; h->core.L = c.num_layers;
    h->core.alpha = 1.0f; h->core.eps = 1e-5f; h->core.hook_site = 0;
    h->core.fast = avxh::cfg_fast(c.residual_dtype);
    h->core.batch_invariant = avxh::cfg_batch_invariant(c.residual_dtype);
    avxh::fold_policy(h->core.fast, c.embed_dim, c.ffn_dim, &h->core.fold, &h->core.fold_min_rows, h->core.batch_invariant);
    h->chunk = c.max_chunk_clips > 0 ? c.max_chunk_clips : 64;       // layer 0 of the extractor holds 32 MB per 10 s clip
    if (h->init_alarm() != AVEXHIP_OK || aves_build(h, tensors, n_tensors) != AVEXHIP_OK || h->weights_fit() != AVEXHIP_OK) { delete h; return nullptr; }
    return h;
}

extern "C" void avexhip_aves_destroy(avexhip_aves* h) { delete h; }

extern "C" int avexhip_aves_num_tokens(const avexhip_aves* h, int64_t T) {
    if (!h) return 0;
    int F[8], P[8];
    return frame_plan(h, T, F, P) == 0 ? F[h->n_conv - 1] : 0;
}

extern "C" size_t avexhip_aves_workspace_bytes(const avexhip_aves* h, int B, int64_t T) {
    if (!h || B <= 0) return 0;
    int F[8], P[8];
    if (frame_plan(h, T, F, P) != 0) return 0;
    return aves_carve(h, nullptr, chunk_for(h->chunk, B, F[h->n_conv - 1]), T, F, P).total;
}

extern "C" int avexhip_aves_forward(avexhip_aves* h, const float* wav, int B, int64_t T, int64_t wav_stride, const uint8_t* frame_pad,
                                    uint32_t hook_mask, float* const* hook_out, int hook_pooled, float* features_out, float* pooled_out,
                                    void* workspace, size_t ws_bytes, void* stream) {
    AVX_REQUIRE(h && wav, "aves_forward: null handle or input");
    AVX_REQUIRE(B > 0 && T > 0, "aves_forward: empty input B=%d T=%lld", B, (long long)T);
    const int E = h->core.E, L = h->core.L, dt = h->dtype, NC = h->n_conv;
    AVX_REQUIRE(hook_mask == 0 || hook_out, "aves_forward: hook_mask set but hook_out is NULL");
    AVX_REQUIRE(L >= 32 || (hook_mask >> L) == 0, "aves_forward: hook_mask has bits beyond layer %d", L - 1);
    for (int i = 0; i < L; ++i) AVX_REQUIRE(!((hook_mask >> i) & 1u) || hook_out[i], "aves_forward: hook %d selected but hook_out[%d] is NULL", i, i);
    int F[8], P[8];
    AVX_REQUIRE(frame_plan(h, T, F, P) == 0, "aves_forward: audio too short for the feature extractor (%lld samples)", (long long)T);
    const int Tt = F[NC - 1];
    hipStream_t s = (hipStream_t)stream;
    if (wav_stride <= 0) wav_stride = T;
    const int chunk = chunk_for(h->chunk, B, Tt);
    const AvesWs need = aves_carve(h, nullptr, chunk, T, F, P);
    if (!workspace || ws_bytes < need.total) {
        avexhip_set_error("aves_forward: workspace too small (%zu bytes given, %zu needed)", ws_bytes, need.total);
        return AVEXHIP_ERR_WORKSPACE;
    }
    Prof prof{h, s};
    int rc;
#define RC(x) do { rc = (x); if (rc != AVEXHIP_OK) return rc; } while (0)
    for (int c0 = 0; c0 < B; c0 += chunk) {
        const int Bc = (B - c0) < chunk ? (B - c0) : chunk;
        const AvesWs w = aves_carve(h, (char*)workspace, chunk, T, F, P);
        const int M = Bc * Tt;
        const double Md = (double)M;
        const bool fast = h->core.fast;
        // 1. feature extractor: layer 0 = Conv1d(1, 512, 10, 5) + GroupNorm over time + GELU; layers 1.. = GEMMs on strided rows
        prof.begin("wavconv0", 2.0 * Bc * F[0] * (double)CC * h->ck[0]);
        RC(avexhip_wavconv0(wav + (size_t)c0 * wav_stride, Bc, T, wav_stride, h->w0, h->gn_w, h->gn_b, 1e-5f, w.stats, w.conv[0], P[0], dt, s));
        AVX_HIP_CHECK(hipMemsetAsync(w.conv[0] + (size_t)Bc * P[0] * CC * 2, 0, (size_t)SLACK * CC * 2, s));
        prof.end();
        const char* cur = w.conv[0];
        for (int l = 1; l < NC; ++l) {
            char* out = w.conv[l & 1];
            avx::GemmArgs g;
            memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
            g.A = cur; g.lda = (int64_t)h->cs[l] * CC; g.W = h->wc[l]; g.ldw = (int64_t)h->ck[l] * CC; g.M = Bc * P[l]; g.N = CC; g.K = h->ck[l] * CC;
            g.bias = h->zero_bias; g.gelu = 1; g.out_half = out; g.ldh = CC;
            static const char* const conv_names[8] = {"gemm.conv0", "gemm.conv1", "gemm.conv2", "gemm.conv3", "gemm.conv4", "gemm.conv5", "gemm.conv6", "gemm.conv7"};
            prof.begin(conv_names[l & 7], 2.0 * Bc * F[l] * (double)CC * h->ck[l] * CC);
            RC(avx::gemm(g, dt, s));
            AVX_HIP_CHECK(hipMemsetAsync(out + (size_t)Bc * P[l] * CC * 2, 0, (size_t)SLACK * CC * 2, s));
            prof.end();
            cur = out;
        }
        // the valid frames of every clip, compacted: [Bc, P, 512] -> [Bc, Tt, 512]
        AVX_HIP_CHECK(hipMemcpy2DAsync(w.feats, (size_t)Tt * CC * 2, cur, (size_t)P[NC - 1] * CC * 2, (size_t)Tt * CC * 2, Bc, hipMemcpyDeviceToDevice, s));
        // 2. feature projection (LayerNorm(512) -> Linear(512, E)), positional conv + residual, encoder LayerNorm
        prof.begin("layernorm", 0.0);
        RC(avx::layernorm(nullptr, w.feats, CC, h->fp_ln_w, h->fp_ln_b, 1e-5f, M, CC, nullptr, CC, w.h0, CC, dt, s));
        prof.end();
        avx::GemmArgs g;
        memset(&g, 0, sizeof(g)); g.ovf = h->d_ovf;
        g.A = w.h0; g.lda = CC; g.W = h->fp_w; g.ldw = CC; g.M = M; g.N = E; g.K = CC; g.bias = h->fp_b;
        g.out_half = w.core.xh; g.ldh = E;
        if (!fast) { g.out_f32 = w.core.x; g.ldo = E; }
        prof.begin("gemm.feature_projection", 2.0 * Md * E * CC);
        RC(avx::gemm(g, dt, s));
        prof.end();
        float* pre32 = fast ? nullptr : w.core.pre;
        void* preh = fast ? w.core.preh : nullptr;
        prof.begin("posconv", 2.0 * Md * E * (E / h->cfg.pos_conv_groups) * h->cfg.pos_conv_kernel);
        RC(avx::posconv(w.core.xh, fast ? nullptr : w.core.x, h->pc_w, h->pc_b, Bc, Tt, E, h->cfg.pos_conv_groups, h->cfg.pos_conv_kernel, pre32, preh, dt, s));
        prof.end();
        prof.begin("layernorm", 0.0);
        RC(avx::layernorm(pre32, preh, E, h->enc_ln_w, h->enc_ln_b, 1e-5f, M, E, fast ? nullptr : w.core.x, E, w.core.xh, E, dt, s));
        prof.end();
        // 3. the layers; hook i = encoder.transformer.layers.{i}.feed_forward.output_dense (aves_model.py:100-126)
        CoreIo io;
        io.Bc = Bc; io.Tt = Tt; io.c0 = (size_t)c0; io.pad = frame_pad ? frame_pad + (size_t)c0 * Tt : nullptr;
        io.hook_mask = hook_mask; io.hook_bit0 = 0; io.hook_out = hook_out; io.hook_pooled = hook_pooled < 0 ? 0 : (hook_pooled > 3 ? 3 : hook_pooled);
        io.features_out = features_out; io.pooled_out = pooled_out;
        RC(avxh::run_layers(h, h->core, h->layers, w.core, io, prof, s));
    }
#undef RC
    { const int rc2 = h->mirror_alarm(s); if (rc2 != AVEXHIP_OK) return rc2; }
    return prof.collect();
}

extern "C" int avexhip_aves_overflow_count(avexhip_aves* h, uint32_t* events, void* sync_stream, int synchronize) {
    AVX_REQUIRE(h && events, "aves_overflow_count: null argument");
    return h->overflow_count(events, (hipStream_t)sync_stream, synchronize);
}
extern "C" int avexhip_aves_set_profiling(avexhip_aves* h, int enabled) {
    AVX_REQUIRE(h, "aves_set_profiling: null handle");
    h->profiling = enabled != 0;
    return AVEXHIP_OK;
}
extern "C" int avexhip_aves_last_profile(const avexhip_aves* h, const char* const** names, const float** ms, const double** flops, int* count) {
    AVX_REQUIRE(h && names && ms && flops && count, "aves_last_profile: null argument");
    *names = h->prof_name_ptrs.data(); *ms = h->prof_ms.data(); *flops = h->prof_flops.data(); *count = (int)h->prof_name_ptrs.size();
    return AVEXHIP_OK;
}
