/*
 * avexhip.h — C ABI of the MI355X (gfx950) embedding-extraction path for the AVEX plugin API.
 *
 * The reference (earthspecies/avex v1.2.0) is 100 % Python and has NO FFI of its own for this
 * path: the boundary it exposes is a Python class contract (ModelBase subclass registered in the
 * model registry).  This library is what a reference-side binding for the hot path would bind
 * (ctypes stub shown in INTEGRATION.md).  Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch/HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the null stream).
 *   - all "dev" pointers are device (HBM) pointers owned by the caller; the library never frees
 *     them and never synchronises the stream (no hidden hipDeviceSynchronize in any *_forward).
 *   - return value: 0 on success, <0 on error; avexhip_last_error() gives a thread-local message.
 *   - "half" buffers hold the operand type chosen at handle creation (AVEXHIP_F16 or AVEXHIP_BF16).
 */
#ifndef AVEXHIP_H
#define AVEXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVEXHIP_ABI_VERSION 9

enum { AVEXHIP_F16 = 0, AVEXHIP_BF16 = 1 };

enum {
    AVEXHIP_OK = 0,
    AVEXHIP_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    AVEXHIP_ERR_HIP = -2,       /* a HIP runtime call failed */
    AVEXHIP_ERR_MISSING = -3,   /* a required tensor is missing from the weight table */
    AVEXHIP_ERR_WORKSPACE = -4  /* workspace too small */
};

const char* avexhip_last_error(void);
int avexhip_abi_version(void);
/* Number of visible HIP devices (does not initialise a device context beyond the count query). */
int avexhip_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Frontend: kaldi-compatible batched log-mel filterbank.
 * Replaces avex/models/beats/beats.py:39-163 (_BatchedFbank) + :304-323 (BEATs.preprocess).
 * One fused kernel: frame (win/hop) -> per-frame DC removal -> pre-emphasis -> window ->
 * zero-pad to 512 -> LDS radix-8 FFT -> |.|^2 -> sparse triangular mel -> log(max(.,eps)) -> affine.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_fbank_plan avexhip_fbank_plan;

typedef struct {
    int32_t win_length;      /* 400  (beats.py:60)   must be <= 512 */
    int32_t hop_length;      /* 160  (beats.py:61) */
    int32_t n_mels;          /* 128 */
    float   input_scale;     /* 32768.0 for BEATs (beats.py:322); 1.0 for EAT */
    float   preemph;         /* 0.97 (beats.py:53) */
    int32_t remove_dc;       /* 1: per-frame mean subtraction (beats.py:140) */
    float   log_floor;       /* 1.1920929e-07 = fp32 eps (beats.py:36,163) */
    float   norm_mean;       /* 15.41663; output = (logmel - norm_mean) / norm_div (beats.py:323) */
    float   norm_div;        /* 2 * 6.55582; use mean 0 / div 1 for raw log-mel */
} avexhip_fbank_config;

/* window: [win_length] fp32, mel_fb: [257, n_mels] fp32 row-major (the reference's persistent
 * buffers backbone.fbank.window / backbone.fbank.mel_fb, beats.py:76,80).  Host or device pointers. */
avexhip_fbank_plan* avexhip_fbank_plan_create(const avexhip_fbank_config* cfg,
                                              const float* window, const float* mel_fb);
void avexhip_fbank_plan_destroy(avexhip_fbank_plan* plan);
/* frames produced for T samples: 1 + (T - win)/hop (snip_edges), 0 if T < win (beats.py:136). */
int avexhip_fbank_num_frames(const avexhip_fbank_plan* plan, int64_t T);
/* wav_dev: [B, T] fp32 (row stride = wav_stride elements).  out_dev: [B, frames, n_mels] fp32. */
int avexhip_fbank_forward(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T,
                          int64_t wav_stride, float* out_dev, void* stream);
/* EAT frontend (avex/models/eat/audio_processor.py:72-143): the same kernel with the per-clip offset `mono - mono.mean()`
 * (:107) subtracted at load, and a fixed number of output rows per clip: rows past the last frame are the zero-padded
 * log-mel rows AFTER normalisation, (0 - norm_mean) / norm_div (:121-135); frames past out_frames are cut.
 * clip_offset_dev: [B] fp32 or NULL.  out_dev: [B, out_frames, n_mels] fp32. */
int avexhip_clip_mean(const float* wav_dev, int B, int64_t T, int64_t wav_stride, float* mean_dev, void* stream);
int avexhip_fbank_forward_padded(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T,
                                 int64_t wav_stride, const float* clip_offset_dev, int out_frames,
                                 float* out_dev, void* stream);

/* The same frontend writing what a 16 x 16 patch-embedding GEMM reads: out_patch_dev [B, out_frames/patch, n_mels/patch, patch*patch]
 * in the operand type (token = t * (n_mels/patch) + f, element = (frame % patch) * patch + (bin % patch)), i.e. Conv2d(1, D, patch,
 * stride patch) over the [frames, n_mels] image becomes avexhip_gemm with K = patch*patch (BEATs beats.py:349-352; EAT's
 * local_encoder, avex/models/eat_hf.py:274).  clip_offset_dev / out_frames as for avexhip_fbank_forward_padded (NULL / 0: none). */
int avexhip_fbank_forward_patches(const avexhip_fbank_plan* plan, const float* wav_dev, int B, int64_t T, int64_t wav_stride,
                                  const float* clip_offset_dev, int out_frames, int patch, void* out_patch_dev, int dtype,
                                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Frontend: STFT power spectrogram / mel spectrogram of the reference's AudioProcessor
 * (avex/data/audio_utils.py:77-172): torch.stft(n_fft, hop, win_length, window, center (reflect pad)) -> |.|^2 ->
 * optional MelScale matrix -> optional _normalize: log(x + 1e-6), then per-clip (x - min) / (max - min + 1e-8).
 * The transform is a mixed-radix (2, 3, 4, 5) Stockham FFT in LDS for n_fft = 2^a 3^b 5^c up to 2048 (the reference's 2048 / 800 /
 * 512 / 400), two frames packed per complex transform; other even n_fft <= 1024 run as a dense fp32-MFMA product with a
 * precomputed, window-folded DFT matrix.  Output [B, n_bins, frames] fp32, n_bins = n_mels or n_fft/2 + 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_melspec_plan avexhip_melspec_plan;
typedef struct {
    int32_t n_fft;        /* even, 64..2048 (prime factors above 5: <= 1024) */
    int32_t hop_length;
    int32_t win_length;   /* <= n_fft; the window is centre-padded to n_fft like torch.stft */
    int32_t n_mels;       /* 0: power spectrogram */
    int32_t center;       /* 1: reflect-pad n_fft/2 on both sides */
    int32_t normalize;    /* 1: log + per-clip min-max (audio_utils.py:166-172) */
} avexhip_melspec_config;
/* window [win_length] fp32; mel_fb [n_fft/2+1, n_mels] fp32 row-major (torchaudio MelScale.fb) or NULL. */
avexhip_melspec_plan* avexhip_melspec_plan_create(const avexhip_melspec_config* cfg, const float* window, const float* mel_fb);
void avexhip_melspec_plan_destroy(avexhip_melspec_plan* plan);
int avexhip_melspec_num_frames(const avexhip_melspec_plan* plan, int64_t T);
int avexhip_melspec_num_bins(const avexhip_melspec_plan* plan);
/* minmax_dev: [B, 2] int32 scratch (needed when normalize = 1). */
int avexhip_melspec_forward(const avexhip_melspec_plan* plan, const float* wav_dev, int B, int64_t T, int64_t wav_stride,
                            float* out_dev, int* minmax_dev, void* stream);

/* EfficientNet-B0 building blocks that are not GEMMs (avex/models/efficientnet.py:55-66 -> torchvision efficientnet_b0).
 * Activations are NHWC half with channels padded to Cp (a multiple of 8; the handle uses 64 or multiples of 128; padding channels stay zero); 1x1 convolutions
 * are avexhip_gemm calls with BatchNorm folded into weight/bias and gelu = 2 (SiLU).
 *  stem:   img [B, H, W] fp32 (the mel image; the reference feeds three copies of it, efficientnet.py:133-135) ->
 *          Conv2d(3->32, 3x3, s2, p1) with channel-summed, BN-folded weights w [9, Cp] + bias [Cp] + SiLU -> [B, Ho, Wo, Cp];
 *          raw_dev (optional, fp32 [B, Ho, Wo, Cp]) receives the pre-activation values.
 *  dwconv: depthwise k x k (3 | 5), stride (1 | 2), padding (k-1)/2, w [k*k, Cp] BN-folded, + SiLU; pool_dev [B, Cp] fp32
 *          (optional) receives the per-clip channel sums of the output (squeeze of squeeze-excitation); that form wants part_dev,
 *          avexhip_effnet_dwconv_part_bytes() bytes of scratch: every workgroup leaves one row of partial sums there and a second
 *          small kernel adds the rows in order, so the sums (and everything after them) repeat bit for bit from run to run.
 *  se:     scale[b, c] = sigmoid(W2 silu(W1 (pool / hw) + b1) + b2), then x[b, :, c] *= scale[b, c] in place (x_dev NULL: the scale
 *          vector only -- the EfficientNet handle applies it inside the projection that follows when that runs in the skinny kernel). */
int avexhip_effnet_stem(const float* img_dev, int B, int H, int W, const float* w_dev, const float* bias_dev, int Cp,
                        void* out_dev, float* raw_dev, int dtype, void* stream);
size_t avexhip_effnet_dwconv_part_bytes(int B, int H, int W, int Cp, int k, int stride);
int avexhip_effnet_dwconv(const void* in_dev, int B, int H, int W, int Cp, int k, int stride, const float* w_dev,
                          const float* bias_dev, void* out_dev, float* pool_dev, float* part_dev, size_t part_bytes, int dtype,
                          void* stream);
int avexhip_effnet_se(const float* pool_dev, int B, int64_t hw, int C, int Cp, int Cs, const float* w1_dev, const float* b1_dev,
                      const float* w2_dev, const float* b2_dev, float* scale_dev, void* x_dev, int dtype, void* stream);

/* Audio ingest: samples as a PCM file holds them -> mono float32 -> another sample rate, on the device.
 *   avexhip_pcm_to_mono_f32  interleaved [frames][channels] samples (sample_format 16 / 24 / 32 / 8 = integer PCM widths, 0 = float32,
 *                            64 = float64) -> [frames] fp32, channels averaged (augmentations.py:269-271, birdset_train_splits.py:184-186)
 *   avexhip_resample_*       torchaudio.transforms.Resample(orig, new) (augmentations.py:274-276): band-limited sinc interpolation,
 *                            Hann window (kaiser_beta <= 0) or Kaiser window, lowpass_filter_width 6 and rolloff 0.99 by default;
 *                            x [B, T] -> out [B, ceil(new * T / orig)] (avexhip_resample_out_length).  Parity unpinned (torchaudio absent). */
typedef struct avexhip_resample_plan avexhip_resample_plan;
avexhip_resample_plan* avexhip_resample_plan_create(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, double kaiser_beta);
/* librosa.resample(y, orig_sr, target_sr, res_type="kaiser_best", scale=True) (birdset_train_splits.py:190-196) = resampy's interpolating
 * resampler: half-window table of a Kaiser-windowed sinc (kaiser_best: num_zeros 64, precision 9, rolloff 0.9475937167399596,
 * beta 14.769656459379492), two wings per output sample with linearly interpolated weights; out [B, ceil(T * new / orig)], the samples past
 * int(T * new / orig) zero (librosa's fix_length), everything divided by sqrt(new / orig) when scale_energy.  The plan works with
 * avexhip_resample_out_length / _forward / _plan_destroy.  Parity unpinned (resampy / librosa absent). */
avexhip_resample_plan* avexhip_resample_interp_plan_create(int orig_freq, int new_freq, int num_zeros, int precision, double rolloff, double kaiser_beta,
                                                           int scale_energy);
void avexhip_resample_plan_destroy(avexhip_resample_plan* plan);
int64_t avexhip_resample_out_length(const avexhip_resample_plan* plan, int64_t T);
int avexhip_resample_forward(const avexhip_resample_plan* plan, const float* x_dev, int B, int64_t T, int64_t x_stride, float* out_dev,
                             int64_t out_stride, void* stream);
int avexhip_pcm_to_mono_f32(const void* raw_dev, int sample_format, int channels, int64_t frames, float* out_dev, void* stream);

/* FLAC decode (the reference reads its .flac samples through torchaudio.load / soundfile: augmentations.py:258-262,
 * tests/samples/animalspeak2/16khz).  avexhip_flac_open parses a whole stream held in host memory -- metadata, frame and
 * subframe headers, the Rice-coded residuals, every CRC-8 / CRC-16 -- and returns NULL (avexhip_last_error) on any violation;
 * avexhip_flac_decode_i32 runs the predictors and the inter-channel decorrelation on the device and writes interleaved
 * [total_samples][channels] int32 samples, bit-exact (STREAMINFO carries the MD5 of the unencoded audio: avexhip_flac_info);
 * left_justify != 0 shifts them to 32-bit full scale so that avexhip_pcm_to_mono_f32(format 32) normalises them like soundfile.
 * 4..32 bits per sample, 1..8 channels, fixed or variable block size; 32-bit stereo with side channels (33-bit samples) refused. */
typedef struct avexhip_flac avexhip_flac;
avexhip_flac* avexhip_flac_open(const uint8_t* data_host, size_t n_bytes);
void avexhip_flac_close(avexhip_flac* h);
int avexhip_flac_info(const avexhip_flac* h, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples, uint8_t* md5_16);
int avexhip_flac_decode_i32(const avexhip_flac* h, int32_t* out_dev, int left_justify, void* stream);

/* First layer of the wav2vec2 / AVES convolutional feature extractor (avex/models/aves_model.py:25-33,86 ->
 * torchaudio wav2vec2 ConvLayerBlock 0, extractor_mode "group_norm", no conv bias):
 *   Conv1d(1, 512, k=10, s=5) -> GroupNorm(512, 512, eps) over time per (clip, channel) -> GELU
 * wav_dev [B, T] fp32; w_dev [512, 10] fp32; gn_w/gn_b [512]; stats_dev: avexhip_wavconv0_stats_floats(B, T) floats of scratch,
 * 8-byte aligned (the clip's 65 second-order moments per 1 024-frame block in fp64 -- the layer is linear, so every channel's mean and
 * variance follow from them -- combined in a fixed order: results are bit-reproducible -- and the 512 (scale, shift) pairs per clip);
 * out_dev [B, frames_pad, 512] half (rows >= frames are zero).  The other six conv layers are strided-row
 * avexhip_gemm calls (A row t of clip b = frames s*t .. s*t+k-1 of the previous layer: lda = s * 512, K = k * 512). */
int avexhip_wavconv0_frames(int64_t T);
int64_t avexhip_wavconv0_stats_floats(int B, int64_t T);
int avexhip_wavconv0(const float* wav_dev, int B, int64_t T, int64_t wav_stride, const float* w_dev,
                     const float* gn_w_dev, const float* gn_b_dev, float eps, float* stats_dev, void* out_dev,
                     int frames_pad, int dtype, void* stream);

/* Probe heads on the device: the forward of the reference's online probes after extract_embeddings(), fp32.
 *   avexhip_layer_mix  base_probes.py:197-206 `_sum`: out[i] = sum_l softmax(layer_weights)_l * taps[l][i] (every weight 1.0 when
 *                      layer_weights_dev is NULL, as the reference does without learned weights); `taps` is a HOST array of L <= 16
 *                      device pointers, each to n floats;
 *   avexhip_dense_f32  nn.Linear with an optional activation (0 none, 1 ReLU, 2 erf-GELU, 3 Tanh) and an optional residual added
 *                      after it: out[m][n] = act(bias[n] + sum_k x[m][k] w[n][k]) + resid[m][n]   (linear_probe.py:44-46,66;
 *                      mlp_probe.py:51-73,91; attention_probe.py:59-86,128-134);
 *   avexhip_mha_f32    the attention core of nn.MultiheadAttention(batch_first=True) in eval (attention_probe.py:128): qkv
 *                      [B*T, 3E] = in_proj output (q | k | v thirds, head h at columns h*E/H..), key_pad optional [B, T] uint8
 *                      (1 = ignore key), out [B*T, E] (to be fed to out_proj).  E/H a multiple of 4, <= 128; T <= 2048.
 *   avexhip_seq_interp_linear  base_probes.py:398-411: taps of different sequence lengths are resampled to the shortest with
 *                      F.interpolate(mode="linear", align_corners=False) along the sequence: in [B, Tin, C] -> out [B, Tout, C]. */
int avexhip_seq_interp_linear(const float* in_dev, int B, int Tin, int C, int Tout, float* out_dev, void* stream);
int avexhip_layer_mix(const float* const* taps, int L, const float* layer_weights_dev, int64_t n, float* out_dev, void* stream);
int avexhip_dense_f32(const float* x_dev, int64_t ldx, const float* w_dev, int64_t ldw, const float* bias_dev, const float* resid_dev,
                      int64_t ldr, int M, int N, int K, int act, float* out_dev, int64_t ldo, void* stream);
int avexhip_mha_f32(const float* qkv_dev, int B, int T, int E, int H, const uint8_t* key_pad_dev, float* out_dev, void* stream);
/* One direction of one nn.LSTM layer (batch first, zero initial state; lstm_probe.py:61-68, 100): xg [B, T, 4H] = x W_ih^T + b_ih + b_hh
 * (gate order i, f, g, o), w_hhT [H, 4H] = W_hh transposed, out[b, t, 0..H) at row stride ldo (2H with a column offset for the
 * reverse direction of a bidirectional layer), reverse != 0 walks t = T-1 .. 0.  1 <= H <= 1024; fp32. */
int avexhip_lstm_layer(const float* xg_dev, const float* w_hhT_dev, int B, int T, int H, int reverse, float* out_dev, int64_t ldo, void* stream);
/* Both directions of a bidirectional layer in ONE launch (the two recurrences are independent and each fills a quarter of the chip at 256
 * clips): forward xg / w_hhT -> out, backward xg_rev / w_hhT_rev -> out_rev (normally out + H), same ldo.  Same arithmetic as two calls. */
int avexhip_lstm_layer_pair(const float* xg_dev, const float* w_hhT_dev, const float* xg_rev_dev, const float* w_hhT_rev_dev, int B, int T, int H,
                            float* out_dev, float* out_rev_dev, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks (exported so every kernel can be parity-tested in isolation through the ABI).
 * `dtype` selects the half operand type of the half buffers.
 * ------------------------------------------------------------------------------------------ */
int avexhip_cast_f32_to_half(const float* in_dev, void* out_dev, int64_t n, int dtype, void* stream);
int avexhip_cast_half_to_f32(const void* in_dev, float* out_dev, int64_t n, int dtype, void* stream);

/* out[m, n] = epilogue( sum_k A[m,k] * W[n,k] )      (torch.nn.Linear layout: W is [N, K])
 *   acc' = acc + bias[n]                               (bias may be NULL)
 *   if out_raw:  out_raw[m,n] = acc'                   (fp32 "hook tap", e.g. fc2 raw output)
 *   if resid / resid_half:  acc' = resid[m,n] * alpha + acc'   (DeepNorm residual, backbone.py:360,372)
 *   if gelu:     acc' = gelu_erf(acc')                 (backbone.py:368)
 *   out_f32 / out_half (either may be NULL) receive acc'.
 * Requirements: N % 128 == 0 (64 with the skinny kernel), K % 64 == 0, all leading dims in elements, 16-byte aligned rows. */
typedef struct {
    const void*  A;  int64_t lda;      /* [M, K] half */
    const void*  W;  int64_t ldw;      /* [N, K] half */
    int32_t M, N, K;
    const float* bias;
    const float* resid; int64_t ldr; float alpha;   /* fp32 residual, or ... */
    const void*  resid_half; int64_t ldrh;          /* ... residual in the operand type (resid == NULL) */
    int32_t gelu;                      /* activation after bias (+ residual): 0 none, 1 exact-erf GELU, 2 SiLU, 3 ReLU,
                                          4 tanh-form GELU (modules.py:177-188), 5 tanh.  The erf GELU is evaluated to 4.5e-7
                                          when an fp32 output is requested and to 6.3e-6 (a tenth of an f16 ulp at |gelu| = 0.1)
                                          when the only output is in the operand type; whichever kernel runs, the same bits */
    float* out_f32;  int64_t ldo;
    void*  out_half; int64_t ldh;
    float* out_raw;  int64_t ldraw;
    int32_t variant;                   /* 0 = auto; 1 = 128-tile, register staging; 3 = 128-tile LDS-DMA;
                                          5 (or 2) = 256-tile half-tile LDS-DMA pipeline, persistent workgroups
                                          (needs N % 256 == 0, K >= 128); 7 = skinny streaming kernel for long thin
                                          products (K, N in {64, 128, 256}, N K <= 32768, half output, bias / activation /
                                          half residual only; auto from 32 768 rows): the whole W in LDS, A rows straight
                                          into MFMA operands; 8 = the full-row residual kernel (N = 768, K % 32 == 0,
                                          K >= 128; half output = half or LayerNorm-folded residual * alpha + acc + bias only):
                                          128 rows x all 768 columns per workgroup, A read once, row statistics finished in the
                                          tile; bit-identical to variant 5 */
    /* LayerNorm folded into the GEMMs around it (all NULL/0 = off; the 256-tile kernel only: N % 256 == 0, K >= 128).
     * A tensor y that is only consumed through LayerNorm (backbone.py:363,374: x = LN(residual * alpha + sublayer))
     * stays raw in the operand type.  The GEMM that produces it writes per-row partial statistics
     * [M][N/64][2] = (sum, sum of squares) of its 64-column segments (stats_out, from the fp32 values);
     * avexhip_ln_rowstats turns them into one (rstd, -mean * rstd) pair per row; the consumers take those pairs:
     *  - A operand = LN(y):  pass A = y, W = W * diag(gamma), bias = b + W beta, ln_s[n] = sum_k W'[n][k],
     *    ln_rows = y's pairs; the epilogue forms rstd * acc + (-mean rstd) * ln_s + bias.
     *  - residual = LN(y):   pass lnr_y = y (ld ldy), lnr_rows = its pairs, gamma, beta:
     *    out = alpha * ((y * rstd - mean rstd) * gamma + beta) + acc + bias.  */
    const float* ln_rows; const float* ln_s;
    const void*  lnr_y; int64_t ldy; const float* lnr_rows;
    const float* lnr_gamma; const float* lnr_beta;
    float* stats_out;
    /* Range alarm of f16 outputs (conversions saturate at +-65504 silently): when non-NULL, the kernel adds to this device
     * counter the number of lanes that rounded at least one |value| > 65504 to an f16 output (a lower bound of the number of
     * clipped elements; 0 = nothing clipped).  bf16 outputs cannot overflow and never count. */
    uint32_t* overflow_count;
    /* ABI 6.  A mean-pooled hook tap without the tap (256-tile kernel): the M rows are clips of pool_rows (>= 64) consecutive rows;
     * every 64-row block writes the column sums of acc + bias -- what out_raw would hold -- over its rows, split at the one clip
     * boundary it can contain: pool_part[ceil(M / 64)][2][N] fp32 (slot 0 = the clip of the block's first row).
     * avexhip_pool_reduce adds each clip's blocks in order and divides by pool_rows.  NULL = off. */
    float* pool_part; int32_t pool_rows;
    int32_t pool_mode;                 /* 0: sums (mean after avexhip_pool_reduce); 1: maxima (avexhip_pool_reduce_mode(.., 1));
                                          2: every clip's first row, written straight to pool_part = [M / pool_rows][N] */
    /* 128-tile kernel (variant 3): scratch of splitk_bytes for split-K, or NULL.  With it a product of <= 64 output tiles and
     * K >= 1024 (one clip's fc2) is split up to 8 ways along K -- fp32 partials [S][M][N] -- and finished by a second kernel that
     * adds the partials in order and applies the epilogue. */
    float* splitk_ws; size_t splitk_bytes;
    /* ABI 9.  The finished row statistics of the output: rows_out[m] = (rstd, -mean * rstd), rstd = 1 / sqrt(var + rows_eps) -- what
     * avexhip_ln_rowstats makes of stats_out, bit for bit -- written by the product itself.  The full-row kernel (variant 8: N = 768, built for the attention
     * output projection, backbone.py:572 + :360-362; never chosen automatically) owns all 768 columns of its 128 rows and finishes the
     * statistics in its epilogue; any other kernel needs stats_out as scratch and avexhip_gemm runs
     * avexhip_ln_rowstats behind it.  `rows_out` must be writable up to M rounded up to even.  NULL = off. */
    float* rows_out; float rows_eps;
} avexhip_gemm_args;
int avexhip_gemm(const avexhip_gemm_args* args, int dtype, void* stream);
/* pool_part as written through avexhip_gemm_args.pool_part -> out[b, n] = mean over clip b's T rows of the raw GEMM output (bit-reproducible). */
int avexhip_pool_reduce(const float* part_dev, int B, int T, int N, float* out_dev, int64_t ldo, void* stream);
int avexhip_pool_reduce_mode(const float* part_dev, int B, int T, int N, float* out_dev, int64_t ldo, int mode, void* stream);   /* mode 1: maximum */
/* stats [M][nseg][2] as written through avexhip_gemm_args.stats_out (nseg = row width / 64, even) -> rows [M][2] =
 * (rstd, -mean * rstd) with rstd = 1 / sqrt(var + eps); segments are added in order (bit-reproducible).  `rows` must be
 * readable up to M rounded up to an even number of rows. */
int avexhip_ln_rowstats(const float* stats_dev, int M, int nseg, float eps, float* rows_dev, void* stream);

/* torch.nn.LayerNorm over the last dim C (C % 4 == 0, C <= 1024), eps inside sqrt; writes fp32
 * and/or half copies; the input is fp32 (in_dev) or the operand type (in_half_dev), exactly one
 * non-NULL.  Replaces beats.py:353, backbone.py:176-177,362,373. */
int avexhip_layernorm(const float* in_dev, const void* in_half_dev, int64_t ld_in, const float* weight,
                      const float* bias, float eps, int M, int C, float* out_f32, int64_t ldo,
                      void* out_half, int64_t ldh, int dtype, void* stream);

/* Gated relative-position-bias attention (backbone.py:494-574) for head_dim 64, T <= 512.
 *   qkv: [B*T, 3*E] half rows (q | k | v, head h at columns h*64..h*64+63 of each third)
 *   bias_tab: [H, 2T-1] fp32 Toeplitz table, entry r <-> relative position (j - i) = r - (T-1)
 *   grep_w: [8, 64], grep_b: [8], grep_a: [H]   (backbone.py:421-422,543-551); gate applied iff
 *   grep_w != NULL.  key_pad: optional [B, T] uint8 (1 = padded key -> -inf, backbone.py:555-558).
 *   out: [B*T, E] half. */
int avexhip_attention(const void* qkv_dev, int B, int T, int H, const float* bias_tab,
                      const float* grep_w, const float* grep_b, const float* grep_a,
                      const uint8_t* key_pad, void* out_dev, int dtype, void* stream);

/* Plain multi-head self-attention for the other head widths (32, 96, 128; 64 too): softmax(q k^T / sqrt(head_dim) [+ -inf on padded
 * keys]) v, what torch.nn.MultiheadAttention computes inside the reference's sequence probes (attention_probe.py:60-70: 768 / 8 heads;
 * transformer_probe.py:66-75).  Same qkv / key_pad / out layout as avexhip_attention with head h at columns h*head_dim of each third. */
int avexhip_attention_hd(const void* qkv_dev, int B, int T, int H, int head_dim, const uint8_t* key_pad, void* out_dev, int dtype,
                         void* stream);

/* Convolutional positional embedding (backbone.py:52-68,172-174): grouped Conv1d(E,E,k=128,
 * pad=64,groups=16) + drop-last + GELU, fused with the residual add:
 *   out[b,t,:] = x[b,t,:] + gelu(conv(x_half)[b,t,:] + bias)     (x = x_f32 if given, else x_half;
 *   out_f32 and/or out_half receive the result)
 * w_packed: E*E/G*K halves, weight-norm already folded, in the kernel's own order [G][64 tap pairs][3][4][48 out][8]
 * (k = tap*48 + c_in = 96 u + 32 v + 8 g4 + e), produced by avexhip_posconv_pack and opaque to callers.
 * Only E/G == 48, K == 128 is built. */
int avexhip_posconv_pack(const float* g_dev, const float* v_dev, int E, int groups, int K,
                         void* w_packed_dev, int dtype, void* stream);
int avexhip_posconv(const void* x_half_dev, const float* x_f32_dev, const void* w_packed_dev,
                    const float* bias_dev, int B, int T, int E, int groups, int K,
                    float* out_f32_dev, void* out_half_dev, int dtype, void* stream);

/* EAT / Data2Vec-multi token assembly (the HF remote model behind avex/models/eat_hf.py:201,274; parity unpinned): per clip, row 0 =
 * class token `extra_tokens`, row 1 + t = patch row t + fixed position t; then LayerNorm(C, eps) (`pre_norm`).
 * patches_half [B * n_patches, C] (operand type), pos [n_patches, C], cls [C] fp32 -> out_half / out_f32 [B * (n_patches + 1), C]. */
int avexhip_token_embed_ln(const void* patches_half, const float* pos, const float* cls, const float* ln_w, const float* ln_b, float eps,
                           int B, int n_patches, int C, void* out_half, float* out_f32, int dtype, void* stream);

/* mean over T: in [B, T, C] fp32 -> out [B, C] fp32 (features.mean(dim=1), README:80). */
int avexhip_mean_pool(const float* in_dev, int B, int T, int C, float* out_dev, void* stream);

#ifdef AVEX_DIAG   /* diagnostic build only (AVEX_AMD_DIAG=1 python -m avex_amd.build -> libavexhip_diag.so); the product library exports none of these */
/* Debug aid: `blocks` workgroups hold a 26 880-byte LDS pattern for `iters` re-check rounds;
 * report_dev[4] (zeroed by the caller) = {mismatches, first bad word, value seen, block}. */
int avexhip_debug_lds_canary(int blocks, int iters, unsigned* report_dev, void* stream);
/* Debug aid: switch on per-workgroup wall-clock stamps (100 MHz) in the 256-tile GEMM and/or copy them out:
 * host_out[4*b + {0,1,2,3}] = start, prologue complete, K loop complete, epilogue stores retired. */
int avexhip_debug_gemm_stamps(int enable, unsigned long long* host_out, int n_blocks);
/* Debug aid: shader-clock counter (s_memtime) at the start and end of each tile's K loop in the persistent GEMM
 * (variant 5), host_out[2*tile + {0,1}]; with the stamps above this gives the clock the chip held in the loop. */
int avexhip_debug_gemm_clocks(unsigned long long* host_out, int n_tiles);
/* Debug aid: s_memtime at the top of every K-tile (and after the last one) of each workgroup's third tile in the persistent GEMM
 * (variant 5, stamps enabled): host_out[64 * block + kt], blocks < 256, kt < 64. */
int avexhip_debug_gemm_kclocks(unsigned long long* host_out, int n_blocks);
#endif /* AVEX_DIAG */

/* T5 bidirectional bucket of a relative position (backbone.py:438-473).  Pure host function. */
int avexhip_rel_bucket(int rel, int num_buckets, int max_distance);

/* ------------------------------------------------------------------------------------------
 * BEATs encoder handle.  Replaces BEATs.extract_features (beats.py:325-382) +
 * TransformerEncoder.extract_features (backbone.py:151-221) + 12 x layer (backbone.py:350-375)
 * + hook capture (base_model.py:77-99) + mean pooling.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_beats avexhip_beats;

typedef struct {
    int32_t input_patch_size;        /* 16 */
    int32_t embed_dim;               /* 512 */
    int32_t encoder_layers;          /* 12 */
    int32_t encoder_embed_dim;       /* 768 */
    int32_t encoder_ffn_embed_dim;   /* 3072 */
    int32_t encoder_attention_heads; /* 12 */
    int32_t conv_pos;                /* 128 */
    int32_t conv_pos_groups;         /* 16 */
    int32_t num_buckets;             /* 320 */
    int32_t max_distance;            /* 800 */
    int32_t gru_rel_pos;             /* 1 */
    int32_t deep_norm;               /* 1: alpha = (2 L)^(1/4) */
    int32_t num_mel_bins;            /* 128 */
    float   sample_frequency;        /* 16000 */
    float   frame_length_ms;         /* 25 */
    float   frame_shift_ms;          /* 10 */
    float   fbank_mean;              /* 15.41663 */
    float   fbank_std;               /* 6.55582 */
    int32_t operand_dtype;           /* AVEXHIP_F16 (default; parity 3e-4) or AVEXHIP_BF16 (2e-3) */
    int32_t max_chunk_clips;         /* clips processed per internal pass (0 = default 256) */
    int32_t residual_dtype;          /* bit 0 -- 0: residual stream + pre-LN sums kept fp32 between kernels (frame-level
                                        error 4e-4); 1: kept in the operand type (1.5e-3 frame level, pooled
                                        unchanged at 2.8e-4, ~25 % less HBM traffic).  Hook taps and the
                                        features output are fp32 either way.
                                        bit 1 (ABI 7) -- AVEXHIP_RESIDUAL_BATCH_INVARIANT, in every handle config that has
                                        this field: a clip's outputs are bit-identical whatever batch it arrives in (the
                                        LayerNorm fold at every batch size, no split-K, one final LayerNorm + pool path);
                                        without it small batches take quicker kernels whose roundings differ in the last
                                        bits (both inside the parity bar).  The reference's fp32 path is batch-independent
                                        (beats_model.py:279-429). */
    /* ABI 6: the rest of BEATsConfig's architecture switches (beats.py:181-196); zero = the official checkpoints' values */
    int32_t layer_norm_first;        /* 0: post-LN blocks (backbone.py:350-375); 1: pre-LN blocks + LayerNorm after the stack
                                        (backbone.py:328-348, 146-147); excludes deep_norm (beats.py:275) */
    int32_t activation_fn;           /* AVEXHIP_FFN_* below (modules.py:203-237) */
    int32_t conv_bias;               /* 1: the patch embedding has a bias ("patch_embedding.bias", beats.py:263-269) */
    /* ABI 8: a rung of the f16 range ladder (see "Range alarm" below).  k > 0: every layer's fc1 stores its hidden activations
       x 2^-k and fc2's weights are packed x 2^k -- both exact powers of two, so the fp32 accumulation of fc2 sees the products
       it would have seen; hidden activations up to 65504 x 2^k fit the f16 operand (the reference computes them in fp32,
       backbone.py:365-370).  The handle then runs the LayerNorm kernels and the generic GEMM epilogues (slower; for the
       retry of a batch whose default forward raised the alarm).  0 = off.  Refused with the GLU feed-forward and when a
       fc2 weight x 2^k leaves the f16 range. */
    int32_t hidden_shift;
} avexhip_beats_config;

#define AVEXHIP_RESIDUAL_BATCH_INVARIANT 2   /* OR into residual_dtype */

/* activation_fn of the config: get_activation_fn's names (modules.py:203-237).  GLU = fc1 replaced by GLU_Linear(E, F, "swish")
 * (backbone.py:296-297): weights "fc1.linear.weight" [2F, E] / "fc1.linear.bias" [2F], hidden = y[:, :F] * swish(y[:, F:]). */
#define AVEXHIP_FFN_GELU 0       /* "gelu": exact erf form (default) */
#define AVEXHIP_FFN_RELU 1       /* "relu" */
#define AVEXHIP_FFN_GELU_TANH 2  /* "gelu_accurate" / "gelu_fast": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) */
#define AVEXHIP_FFN_TANH 3       /* "tanh" */
#define AVEXHIP_FFN_LINEAR 4     /* "linear" */
#define AVEXHIP_FFN_GLU 5        /* "glu" */

/* Weight table entry: reference state-dict key ("backbone." prefix optional), fp32 data
 * (host or device pointer), element count.  (avex/models/utils/load.py:521-570 ingest.) */
typedef struct {
    const char*  name;
    const float* data;
    int64_t      numel;
} avexhip_tensor;

avexhip_beats* avexhip_beats_create(const avexhip_beats_config* cfg, const avexhip_tensor* tensors,
                                    int n_tensors);
void avexhip_beats_destroy(avexhip_beats* h);

/* Tokens produced for T samples: (frames/16) * (mel/16)  (beats.py:350-352). */
int avexhip_beats_num_tokens(const avexhip_beats* h, int64_t T);
/* Bytes of caller-provided workspace needed for a forward of B clips of T samples. */
size_t avexhip_beats_workspace_bytes(const avexhip_beats* h, int B, int64_t T);

/* hook_mask bit i (i = 0..L) selects layer i of the reference's layer map
 * (0 = backbone.post_extract_proj, i = backbone.encoder.layers.{i-1}.fc2, beats_model.py:206-227).
 * hook_out[i] receives the raw module output, batch-first [B, T', E] fp32 (hook_pooled == 0) or, reduced over
 * the T' tokens to [B, E], its mean (1), maximum (2) or first token (3): extract_embeddings' aggregations
 * (beats_model.py:403-417); clips of >= 64 tokens are reduced inside the GEMM epilogue that produces the tap
 * (no [B, T', E] tensor exists).  features_out: [B, T', E] fp32 or NULL.
 * pooled_out: [B, E] fp32 (features.mean(dim=1)) or NULL.  frame_pad: optional [B, T'] uint8
 * token padding mask (1 = padded; beats.py:283-302 geometry is applied by the caller). */
int avexhip_beats_forward(avexhip_beats* h, const float* wav_dev, int B, int64_t T,
                          int64_t wav_stride, const uint8_t* frame_pad, uint32_t hook_mask,
                          float* const* hook_out, int hook_pooled, float* features_out,
                          float* pooled_out, void* workspace, size_t workspace_bytes, void* stream);

/* Same, but starting from normalised fbank features [B, frames, n_mels] fp32 (skips the frontend). */
int avexhip_beats_forward_fbank(avexhip_beats* h, const float* fbank_dev, int B, int frames,
                                const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out,
                                int hook_pooled, float* features_out, float* pooled_out,
                                void* workspace, size_t workspace_bytes, void* stream);

/* A forward recorded as a hipGraph (ABI 6).  At small batch the path is launch-bound (about 95 kernels of a few microseconds each for
 * one clip); graph_capture runs the forward described by its arguments once (so that nothing is created lazily afterwards), records
 * the same forward from `stream` (not the NULL stream) with hipStreamBeginCapture / EndCapture and instantiates it; graph_launch
 * replays it on any stream.
 * Every pointer is baked into the graph: the caller keeps wav_dev, the outputs, frame_pad and the workspace alive at the same
 * addresses and refills wav_dev in place between launches.  The handle must not be in profiling mode; the recorded forward runs on
 * one stream whatever AVEX_AMD_STREAMS says.  The range alarm's host mirror is part of the graph.  Returns NULL on error. */
typedef struct avexhip_beats_graph avexhip_beats_graph;
avexhip_beats_graph* avexhip_beats_graph_capture(avexhip_beats* h, const float* wav_dev, int B, int64_t T, int64_t wav_stride,
                                                 const uint8_t* frame_pad, uint32_t hook_mask, float* const* hook_out,
                                                 int hook_pooled, float* features_out, float* pooled_out, void* workspace,
                                                 size_t workspace_bytes, void* stream);
int avexhip_beats_graph_launch(avexhip_beats_graph* g, void* stream);
int avexhip_beats_graph_nodes(const avexhip_beats_graph* g);      /* kernel / copy nodes recorded */
void avexhip_beats_graph_destroy(avexhip_beats_graph* g);

/* Per-stage timing of the most recent forward on this handle is available when the handle was put
 * in profiling mode (HIP events on the launch stream; forces a stream sync at the end of forward).
 * names/ms arrays are library-owned and valid until the next forward. */
/* Range alarm.  The residual stream and every GEMM output of the default mode are f16; conversions saturate at +-65504 instead of
 * producing inf, silently.  Each handle keeps a sticky count of the lanes that clipped at least one value (see
 * avexhip_gemm_args.overflow_count); every forward ends with an asynchronous copy of it to pinned host memory.
 * overflow_count returns that host copy: with synchronize = 0 whatever has arrived (never blocks: a forward still in flight is
 * not included), with synchronize != 0 after hipStreamSynchronize(sync_stream).  0 = nothing was ever clipped; > 0 = results
 * since the last reset may be wrong: rerun with operand_dtype bf16 (fp32's exponent range) and residual_dtype 0 (fp32 stream).
 * bf16 handles never count.  The reference computes in fp32 throughout (backbone.py:350-375). */
int avexhip_beats_overflow_count(avexhip_beats* h, uint32_t* events, void* sync_stream, int synchronize);
int avexhip_beats_overflow_reset(avexhip_beats* h, void* stream);

int avexhip_beats_set_profiling(avexhip_beats* h, int enabled);
int avexhip_beats_last_profile(const avexhip_beats* h, const char* const** names, const float** ms,
                               const double** flops, int* count);

/* ------------------------------------------------------------------------------------------
 * A stack of post-LN transformer layers on caller-provided token rows: what torch.nn.TransformerEncoder (norm_first = False,
 * batch_first = True) does inside the reference's TransformerProbe (avex/models/probes/transformer_probe.py:65-72, 112), on the half-
 * precision kernels of the encoders (f16 / bf16 operands, fp32 accumulate: 1e-3 of the fp32 module, ten times its speed).
 * Weight table keys: layers.{i}.self_attn.in_proj.{weight,bias} (PyTorch's in_proj_weight / in_proj_bias under these names),
 * layers.{i}.self_attn.out_proj.*, layers.{i}.norm1.*, layers.{i}.linear1.*, layers.{i}.linear2.*, layers.{i}.norm2.*.
 * Head width embed_dim / num_heads = 32, 64, 96 or 128 (the probe configs the reference ships use 8 heads of 96), widths multiples of
 * 128.  ffn_dim = 0: attention-only blocks x = norm1(x + attn(x)) -- the AttentionProbe's layers (attention_probe.py:60-70, 127-130),
 * whose parameters go under the same layers.{i}.self_attn.* / layers.{i}.norm1.* names.  x [B, T, E] fp32; key_pad [B, T] (1 = masked
 * key) or NULL; features_out [B, T, E] fp32 (the last norm) and / or pooled_out [B, E] (its plain mean over the T rows).
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_stack avexhip_stack;
typedef struct {
    int32_t embed_dim;        /* 768 */
    int32_t num_heads;        /* 12 (head width 64) */
    int32_t num_layers;       /* 4 */
    int32_t ffn_dim;          /* dim_feedforward (the probe's attention_dim: 768); 0 = no feed-forward half */
    float   norm_eps;         /* 1e-5 */
    int32_t activation;       /* avexhip_gemm_args.gelu code of the feed-forward: 3 ReLU (PyTorch's default), 1 erf GELU */
    int32_t operand_dtype;
    int32_t max_chunk_clips;  /* 0 = 256 (scaled down for sequences longer than 512) */
    int32_t residual_dtype;   /* as avexhip_beats_config */
} avexhip_stack_config;
avexhip_stack* avexhip_stack_create(const avexhip_stack_config* cfg, const avexhip_tensor* tensors, int n_tensors);
void avexhip_stack_destroy(avexhip_stack* h);
size_t avexhip_stack_workspace_bytes(const avexhip_stack* h, int B, int T);
int avexhip_stack_forward(avexhip_stack* h, const float* x_dev, int B, int T, const uint8_t* key_pad, float* features_out, float* pooled_out,
                          void* workspace, size_t workspace_bytes, void* stream);
int avexhip_stack_overflow_count(avexhip_stack* h, uint32_t* events, void* sync_stream, int synchronize);

/* ------------------------------------------------------------------------------------------
 * EAT encoder handle: what EATHFModel.forward does per batch (avex/models/eat_hf.py:241-289) --
 * EATAudioProcessor's kaldi fbank image (avex/models/eat/audio_processor.py:72-143) and the HF remote Data2Vec-multi image
 * encoder (backbone.extract_features): 16 x 16 patch rows -> local_encoder -> class token + fixed positions + pre_norm ->
 * `depth` post-LN blocks -> features [B, n_patches + 1, embed_dim].  Same contract as the BEATs handle: the weight table is
 * copied at creation (keys as in the reference's state dict, with or without the "backbone." / "backbone.model." prefixes:
 * local_encoder.proj.*, extra_tokens, fixed_positional_encoder.positions, pre_norm.*, blocks.{i}.attn.qkv / attn.proj / norm1 /
 * mlp.fc1 / mlp.fc2 / norm2), the caller owns every buffer and the stream, nothing synchronises.
 * PARITY UNPINNED: the remote model code is absent from the reference tree; checker oracle/eat_oracle.py.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_eat avexhip_eat;
typedef struct {
    int32_t embed_dim;        /* 768 */
    int32_t num_heads;        /* 12 (head width 64) */
    int32_t depth;            /* 12 */
    int32_t ffn_dim;          /* 3072 */
    int32_t patch_size;       /* 16 */
    int32_t target_length;    /* 1024 frames (eat_hf.py:150) */
    int32_t n_mels;           /* 128 */
    float   norm_eps;         /* 1e-6 */
    float   norm_mean;        /* -4.268: the image is (logmel - norm_mean) / (2 norm_std) (audio_processor.py:137) */
    float   norm_std;         /* 4.569 */
    int32_t operand_dtype;    /* AVEXHIP_F16 / AVEXHIP_BF16 */
    int32_t max_chunk_clips;  /* 0 = 256 */
    int32_t residual_dtype;   /* 0 = fp32 residual stream, 1 = operand type (LayerNorms folded into the GEMMs) */
} avexhip_eat_config;
avexhip_eat* avexhip_eat_create(const avexhip_eat_config* cfg, const avexhip_tensor* tensors, int n_tensors);
void avexhip_eat_destroy(avexhip_eat* h);
int avexhip_eat_num_tokens(const avexhip_eat* h);                  /* n_patches + 1 (513) */
size_t avexhip_eat_workspace_bytes(const avexhip_eat* h, int B);
/* Exactly one of wav_dev ([B, T] fp32, row stride wav_stride) / spec_dev ([B, target_length, n_mels] fp32, an EATAudioProcessor image).
 * hook_mask bit i selects blocks.{i}.attn.proj's raw output (eat_hf.py:220-236) -> hook_out[i] ([B, tokens, E], or [B, E] token
 * means when hook_pooled).  features_out [B, tokens, E] (the last norm2) and / or pooled_out [B, E] with pooling 1 = class token,
 * 2 = mean over the tokens (eat_hf.py:283-288); pooling 0 <=> pooled_out NULL. */
int avexhip_eat_forward(avexhip_eat* h, const float* wav_dev, int B, int64_t T, int64_t wav_stride, const float* spec_dev,
                        uint32_t hook_mask, float* const* hook_out, int hook_pooled, float* features_out, float* pooled_out,
                        int pooling, void* workspace, size_t workspace_bytes, void* stream);
int avexhip_eat_overflow_count(avexhip_eat* h, uint32_t* events, void* sync_stream, int synchronize);
int avexhip_eat_set_profiling(avexhip_eat* h, int enabled);
int avexhip_eat_last_profile(const avexhip_eat* h, const char* const** names, const float** ms, const double** flops, int* count);

/* ------------------------------------------------------------------------------------------
 * AVES encoder handle: aves_model.Model.forward (avex/models/aves_model.py:128-151) = torchaudio wav2vec2_model(...)
 * .extract_features(x)[0][-1]: convolutional feature extractor (conv_kernel / conv_stride per layer, 512 channels, layer 0
 * with GroupNorm), LayerNorm(512) -> Linear(512, E), weight-normed positional conv, `num_layers` post-LN layers ->
 * features [B, T', E].  Weight table keys as torchaudio names them (with or without the "model." prefix).
 * PARITY UNPINNED: torchaudio is absent from the reference tree; checker oracle/aves_oracle.py.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_aves avexhip_aves;
typedef struct {
    int32_t embed_dim;        /* 768 */
    int32_t num_heads;        /* 12 */
    int32_t num_layers;       /* 12 */
    int32_t ffn_dim;          /* 3072 */
    int32_t pos_conv_kernel;  /* 128 */
    int32_t pos_conv_groups;  /* 16 */
    int32_t n_conv_layers;    /* 7 */
    int32_t conv_kernel[8];   /* 10, 3, 3, 3, 3, 2, 2 (aves_model.py:25-33) */
    int32_t conv_stride[8];   /* 5, 2, 2, 2, 2, 2, 2 */
    int32_t operand_dtype;
    int32_t max_chunk_clips;  /* 0 = 64 (layer 0 of the extractor holds 32 MB per 10 s clip) */
    int32_t residual_dtype;
} avexhip_aves_config;
avexhip_aves* avexhip_aves_create(const avexhip_aves_config* cfg, const avexhip_tensor* tensors, int n_tensors);
void avexhip_aves_destroy(avexhip_aves* h);
int avexhip_aves_num_tokens(const avexhip_aves* h, int64_t T);     /* frames of the last conv layer for T samples, 0 if too short */
size_t avexhip_aves_workspace_bytes(const avexhip_aves* h, int B, int64_t T);
/* hook_mask bit i selects encoder.transformer.layers.{i}.feed_forward.output_dense's raw output (aves_model.py:100-126);
 * frame_pad optional [B, T'] uint8 (1 = padded frame: masked as a key); pooled_out = mean over the T' frames. */
int avexhip_aves_forward(avexhip_aves* h, const float* wav_dev, int B, int64_t T, int64_t wav_stride, const uint8_t* frame_pad,
                         uint32_t hook_mask, float* const* hook_out, int hook_pooled, float* features_out, float* pooled_out,
                         void* workspace, size_t workspace_bytes, void* stream);
int avexhip_aves_overflow_count(avexhip_aves* h, uint32_t* events, void* sync_stream, int synchronize);
int avexhip_aves_set_profiling(avexhip_aves* h, int enabled);
int avexhip_aves_last_profile(const avexhip_aves* h, const char* const** names, const float** ms, const double** flops, int* count);

/* ------------------------------------------------------------------------------------------
 * EfficientNet-B0 / B1 encoder handle: torchvision efficientnet_b0().features as efficientnet.Model.forward calls it
 * (avex/models/efficientnet.py:55-66,163-215) on the AudioProcessor's mel image ([B, n_mels, frames] fp32, made by
 * avexhip_melspec_forward; the reference repeats it to three channels, :138-140).  `stage` rows are torchvision's MBConv settings
 * (expand_ratio, kernel, stride, in_channels, out_channels, repeats): B0 = (1,3,1,32,16,1) (6,3,2,16,24,2) (6,5,2,24,40,2)
 * (6,3,2,40,80,3) (6,5,1,80,112,3) (6,5,2,112,192,4) (6,3,1,192,320,1); B1 adds a repeat per stage as torchvision does.
 * Weight table keys as torchvision names them (with or without the wrapper's "model." prefix): features.0.0 / .0.1 (stem conv + BN),
 * features.{s}.{j}.block.{d}.0 / .1, .fc1 / .fc2 (squeeze-excitation), features.{last}.0 / .1 (head).
 * Hook taps (efficientnet.py:82-114): 0 = model.features.0.0, then every *.block.3.0 in order, last = the head conv; each is the
 * convolution's output BEFORE its BatchNorm, NCHW fp32 [B, C, H', W'] (avexhip_effnet_tap_shape gives C, H', W').
 * PARITY UNPINNED: torchvision is absent from the reference tree; checker oracle/effnet_oracle.py.
 * ------------------------------------------------------------------------------------------ */
typedef struct avexhip_effnet avexhip_effnet;
typedef struct {
    int32_t n_stages;         /* 7 */
    int32_t stage[8][6];      /* expand_ratio, kernel (3 | 5), stride (1 | 2), in_channels, out_channels, repeats */
    int32_t stem_channels;    /* 32 */
    int32_t head_channels;    /* 1280 */
    float   bn_eps;           /* 1e-5 */
    int32_t operand_dtype;
    int32_t max_chunk_clips;  /* 0 = 256 */
} avexhip_effnet_config;
avexhip_effnet* avexhip_effnet_create(const avexhip_effnet_config* cfg, const avexhip_tensor* tensors, int n_tensors);
void avexhip_effnet_destroy(avexhip_effnet* h);
int avexhip_effnet_num_taps(const avexhip_effnet* h);                      /* 17 for B0 */
/* tap -1 = the feature map [head_channels, H', W'] */
int avexhip_effnet_tap_shape(const avexhip_effnet* h, int tap, int H, int W, int* C, int* Ho, int* Wo);
size_t avexhip_effnet_workspace_bytes(const avexhip_effnet* h, int B, int H, int W);
/* mel_dev [B, H, W] fp32; features_out [B, head, H', W'] (efficientnet.py:208) and / or pooled_out [B, head] = mean over H' x W'. */
int avexhip_effnet_forward(avexhip_effnet* h, const float* mel_dev, int B, int H, int W, uint32_t hook_mask, float* const* hook_out,
                           float* features_out, float* pooled_out, void* workspace, size_t workspace_bytes, void* stream);
int avexhip_effnet_overflow_count(avexhip_effnet* h, uint32_t* events, void* sync_stream, int synchronize);
int avexhip_effnet_set_profiling(avexhip_effnet* h, int enabled);
int avexhip_effnet_last_profile(const avexhip_effnet* h, const char* const** names, const float** ms, const double** flops, int* count);

#ifdef __cplusplus
}
#endif
#endif /* AVEXHIP_H */
