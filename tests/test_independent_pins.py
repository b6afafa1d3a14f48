"""Second opinions for the oracle rows the reference cannot pin here.

Rows a14 (kaldi fbank with a Hann window), a16 (torchaudio MelScale) and a18 (torchaudio wav2vec2) call into torchaudio, which is
neither under /root/reference nor installed in this image, so their oracles carry "parity unpinned" against the REFERENCE.  The
`transformers` package that IS installed carries independent implementations of the same three algorithms, each written to be
interchangeable with torchaudio's:

* `ASTFeatureExtractor._extract_fbank_features` falls back, when torchaudio is missing, to `audio_utils.spectrogram(...)` with
  kaldi mel filters as the replacement of `torchaudio.compliance.kaldi.fbank(window_type="hanning", num_mel_bins=128)` -- the call
  of avex/models/eat/audio_processor.py:110-119;
* `audio_utils.mel_filter_bank(norm=None, mel_scale="htk")` is the HTK triangular bank of `torchaudio.functional.melscale_fbanks`
  that `torchaudio.transforms.MelScale` (avex/data/audio_utils.py:97-101) builds;
* `Wav2Vec2Model` is the architecture `torchaudio.models.wav2vec2_model` (avex/models/aves_model.py:86-91) implements; torchaudio's
  own `import_huggingface_model` maps one onto the other key by key, which is the key map used below.

These are not the reference (the rows stay "unpinned" in DESIGN.md), but an oracle that agrees with a second, unrelated
implementation to fp32 rounding is not a private reading of the documentation.  CPU only; skipped when transformers is missing.
"""
import numpy as np
import pytest

from avex_amd import synth
from oracle import aves_oracle as AO
from oracle import beats_oracle as O

pytest.importorskip("transformers")


def test_kaldi_hann_fbank_matches_transformers_numpy_fallback():
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    rng = np.random.default_rng(5)
    wav = (0.1 * rng.standard_normal(16000 * 2)).astype(np.float32)
    wav = wav - wav.mean(dtype=np.float32)                                   # audio_processor.py:107
    mel_filters = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=8000,
                                  sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    window = window_function(400, "hann", periodic=False)
    theirs = spectrogram(wav, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97,
                         mel_filters=mel_filters, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    ours = O.fbank(wav[None], n_mels=128, win_length=400, hop_length=160, window=O.hann_window(400))[0]
    assert ours.shape == theirs.shape == (198, 128)
    # log-mel values are O(1..10); transformers computes in float64, the reference path in fp32
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-4)
    # and through the whole EAT frontend: pad to 1024 frames, fixed normalisation
    full = O.eat_preprocess(wav[None])[0]
    ref = np.zeros((1024, 128), np.float32); ref[:198] = theirs
    ref = (ref - np.float32(-4.268)) / (np.float32(4.569) * 2)
    np.testing.assert_allclose(full, ref, rtol=0, atol=5e-5)


@pytest.mark.parametrize("n_fft,n_mels,sr", [(800, 128, 16000), (2048, 128, 16000), (1024, 64, 32000), (512, 40, 8000)])
def test_htk_mel_bank_matches_transformers(n_fft, n_mels, sr):
    from transformers.audio_utils import mel_filter_bank
    theirs = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=sr // 2,
                             sampling_rate=sr, norm=None, mel_scale="htk")
    ours = O.htk_mel_fb(n_fft // 2 + 1, n_mels, sr)
    assert ours.shape == theirs.shape
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-6)


def _hf_wav2vec2(cfg, sd):
    import torch
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    convs = [tuple(int(v) for v in c) for c in cfg["extractor_conv_layer_config"]]
    hc = Wav2Vec2Config(
        hidden_size=int(cfg["encoder_embed_dim"]), num_hidden_layers=int(cfg["encoder_num_layers"]),
        num_attention_heads=int(cfg["encoder_num_heads"]), intermediate_size=int(cfg["encoder_ff_interm_features"]),
        hidden_act="gelu", feat_extract_norm="group", feat_extract_activation="gelu", conv_dim=[c[0] for c in convs],
        conv_kernel=[c[1] for c in convs], conv_stride=[c[2] for c in convs], conv_bias=False,
        num_conv_pos_embeddings=int(cfg["encoder_pos_conv_kernel"]), num_conv_pos_embedding_groups=int(cfg["encoder_pos_conv_groups"]),
        do_stable_layer_norm=False, layer_norm_eps=1e-5, hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        feat_proj_dropout=0.0, layerdrop=0.0, apply_spec_augment=False, attn_implementation="eager")
    model = Wav2Vec2Model(hc).eval()
    # torchaudio.models.wav2vec2.utils.import_huggingface_model's correspondence, inverted
    mapped = {}
    for k, v in sd.items():
        k = k[len("model."):]
        k = k.replace("encoder.feature_projection.", "feature_projection.")
        k = k.replace("encoder.transformer.", "encoder.")
        mapped[k] = torch.from_numpy(np.ascontiguousarray(v))
    own = model.state_dict()
    missing = [k for k in own if k not in mapped and "masked_spec_embed" not in k]
    extra = [k for k in mapped if k not in own]
    assert not missing and not extra, (missing[:5], extra[:5])
    model.load_state_dict(mapped, strict=False)
    return model


@pytest.mark.parametrize("layers,samples", [(2, 8000), (12, 16000)])
def test_aves_oracle_matches_transformers_wav2vec2(layers, samples):
    import torch
    cfg = dict(synth.AVES_BASE_CFG, encoder_num_layers=layers)
    sd = synth.aves_state_dict(cfg, seed=0)
    wav = synth.noise_clips(2, samples, seed=4)
    ours, _taps = AO.aves_forward(wav, sd, cfg)
    model = _hf_wav2vec2(cfg, sd)
    torch.set_num_threads(4)
    with torch.no_grad():
        theirs = model(torch.from_numpy(wav)).last_hidden_state.numpy()
    assert ours.shape == theirs.shape
    rel = np.linalg.norm(ours - theirs) / np.linalg.norm(theirs)
    assert rel < 2e-5, rel
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-4)


@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (22050, 16000), (8000, 16000), (16000, 22050)])
def test_sinc_resampler_matches_scipy_polyphase_engine(orig, new):
    """`oracle/ingest_oracle.resample` (torchaudio's `Resample`: a bank of `new` phase kernels applied with stride `orig`) against an
    unrelated polyphase engine, `scipy.signal.resample_poly`, driven with torchaudio's published prototype filter sampled on the fine
    grid of step 1 / (orig * new): tap j of phase p is prototype sample (j - width) * new - p * orig.  Checks the phase / stride /
    offset / length bookkeeping and the edges; the filter formula itself is common to both sides."""
    import math
    from scipy.signal import resample_poly
    from oracle import ingest_oracle as IO
    g = math.gcd(orig, new)
    o, n = orig // g, new // g
    lpw, rolloff = 6, 0.99
    base = min(o, n) * rolloff
    width = math.ceil(lpw * o / base)
    M = (width + o) * n
    t = np.clip(np.arange(-M, M + 1, dtype=np.float64) / (o * n) * base, -lpw, lpw)
    proto = np.sinc(t) * np.cos(t * math.pi / lpw / 2) ** 2 * (base / o)
    x = np.random.default_rng(orig).standard_normal(5003).astype(np.float32)
    theirs = resample_poly(x.astype(np.float64), n, o, window=proto / n)          # scipy scales an explicit filter by `up`
    ours = IO.resample(x, orig, new)
    assert ours.shape == theirs.shape == (math.ceil(n * 5003 / o),)
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-6)
