"""Pin the CPU oracle (oracle/beats_oracle.py) to the REAL reference.

tests/golden/*.npz were produced by running earthspecies/avex itself (make_goldens.py) on the
synthetic checkpoint; these tests are the oracle's licence to act as the checker for the HIP path.
Also checks the known-answer values of SURVEY.md Appendix B (captured from the reference).
"""
import json
import math

import numpy as np
import pytest

from _util import max_abs, rel_l2
from avex_amd import synth
from oracle import beats_oracle as O


@pytest.fixture(scope="module")
def fb(golden_dir):
    return np.load(f"{golden_dir}/fbank.npz")


def test_window_and_mel_known_answers(fb):
    w = O.povey_window()
    assert max_abs(w, fb["window"]) < 5e-7
    assert w[0] == 0 and abs(w[1] - 2.6513456e-4) < 1e-10 and abs(w.sum() - 212.14700) < 1e-3
    mel = O.mel_filterbank()
    ref = np.zeros((257, 128), np.float32)
    ref[fb["mel_nz_rows"], fb["mel_nz_cols"]] = fb["mel_nz_vals"]
    assert mel.shape == (257, 128) and (mel != 0).sum() == 504
    assert np.array_equal(mel != 0, ref != 0)
    assert max_abs(mel, ref) < 5e-5                      # fp32 log() differs by an ulp between libms
    assert abs(mel.sum() - 252.61052) < 2e-3 and not mel[0].any() and not mel[256].any()
    assert abs(mel[1, 0] - 0.802637) < 1e-4 and abs(mel[57, 63] - 0.947644) < 1e-4


def test_fbank_matches_reference(fb):
    y = O.fbank(synth.noise_clips(2, 16000, seed=0) * np.float32(2 ** 15))
    assert y.shape == (2, 98, 128)
    assert max_abs(y, fb["noise16k"]) < 5e-4            # the reference's own test tolerance is 1e-4 rel+abs
    y = O.fbank(synth.noise_clips(2, 160000, seed=0) * np.float32(2 ** 15))
    assert y.shape == (2, 998, 128)
    assert max_abs(y[:, ::37], fb["noise160k_rows37"]) < 5e-4
    for nm in (64, 256):
        y = O.fbank(synth.noise_clips(1, 4000, seed=7) * np.float32(2 ** 15), n_mels=nm, mel_fb=O.mel_filterbank(512, nm))
        assert y.shape == (1, 23, nm) and max_abs(y, fb[f"noise4k_mel{nm}"]) < 5e-4


def test_fbank_known_answers():
    floor = math.log(1.1920929e-07)
    for wav in (np.zeros((1, 16000), np.float32), np.full((1, 16000), 0.25, np.float32)):
        y = O.fbank(wav * np.float32(2 ** 15))
        assert y.shape == (1, 98, 128) and np.allclose(y, floor, atol=1e-5)     # silence and DC -> log(eps)
    n = np.arange(16000)
    y = O.fbank((0.8 * np.sin(2 * np.pi * 440 * n / 16000)).astype(np.float32)[None] * np.float32(2 ** 15))
    assert int(y[0, 0].argmax()) == 23 and abs(float(y[0, 0].max()) - 25.765612) < 2e-3
    assert abs(float(y[0, 0, 20]) - 20.55091) < 2e-3
    for T, frames in ((4000, 23), (8000, 48), (16000, 98), (32000, 198), (80000, 498), (399, 0)):
        assert O.fbank(np.zeros((1, T), np.float32)).shape[1] == frames


def test_bucket_function_matches_reference(golden_dir):
    g = np.load(f"{golden_dir}/base_api.npz")["bucket_rel_-495..495"]
    mine = O.relative_position_bucket(np.arange(-495, 496)[None, :], 320, 800)[0]
    assert np.array_equal(mine, g)
    for d, up, dn in ((0, 0, 0), (1, 161, 1), (79, 239, 79), (80, 240, 80), (100, 247, 87), (159, 263, 103),
                      (160, 264, 104), (200, 271, 111), (300, 285, 125), (400, 295, 135), (495, 303, 143)):
        assert mine[495 + d] == up and mine[495 - d] == dn
    assert mine.max() == 303 and len(np.unique(mine)) == 287


def test_tiny_config_every_stage(golden_dir):
    ts = np.load(f"{golden_dir}/tiny_stages.npz")
    sd = synth.beats_state_dict(synth.BEATS_TINY_CFG, seed=1)
    x = synth.noise_clips(2, 16000, seed=3)
    f, taps = O.beats_forward(x, sd, synth.BEATS_TINY_CFG, return_stages=True)
    assert f.shape == ts["features"].shape == (2, 48, 64)
    assert rel_l2(f, ts["features"]) < 1e-5
    checked = 0
    for k in ts.files:
        if k in taps:
            assert rel_l2(taps[k], ts[k]) < 1e-5, k
            checked += 1
    assert checked >= 10
    pm = np.zeros((2, 16000), bool); pm[1, 8000:] = True
    fm, _ = O.beats_forward(x, sd, synth.BEATS_TINY_CFG, padding_mask=pm)
    assert rel_l2(fm, ts["features_masked"]) < 1e-5
    frames = 98
    assert np.array_equal(O.forward_padding_mask(48, O.forward_padding_mask(frames, pm)), ts["frame_mask"])


@pytest.fixture(scope="module")
def base(golden_dir):
    g = np.load(f"{golden_dir}/base_api.npz")
    with open(f"{golden_dir}/base_api.json") as f:
        meta = json.load(f)
    return g, meta, synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)


def test_base_pooled_and_hooks(base):
    g, meta, sd = base
    cfg = synth.BEATS_BASE_CFG
    names = O.layer_names(cfg)
    assert names == [meta["layer_map"][str(i)] for i in range(13)]
    for tag, B, T in (("b1", 1, 160000), ("short", 3, 16000)):
        f, taps = O.beats_forward(synth.noise_clips(B, T, seed=0), sd, cfg)
        assert rel_l2(O.pooled(f), g[f"{tag}.pooled"]) < 1e-5
        assert rel_l2(f[:, ::16], g[f"{tag}.feat_tok16"]) < 1e-5
        am = np.concatenate([taps[n].mean(1) for n in names], 1)
        assert rel_l2(am, g[f"{tag}.all_mean"]) < 1e-5


def test_base_aggregations_and_mask(base):
    g, meta, sd = base
    cfg = synth.BEATS_BASE_CFG
    x = synth.noise_clips(2, 32000, seed=5)
    hooks = meta["resolve_[0,-1]"]
    for agg in ("mean", "max", "cls_token"):
        e = O.extract_embeddings(x, sd, cfg, hooks, agg)
        assert e.shape == (2, 1536) and rel_l2(e, g[f"agg.{agg}"]) < 1e-5
    lst = O.extract_embeddings(x, sd, cfg, hooks, "none")
    assert [list(t.shape) for t in lst] == meta["agg_none_shapes"]
    assert rel_l2(lst[1][:, ::8], g["agg.none1_tok8"]) < 1e-5
    pm = np.zeros((2, 32000), bool); pm[1, 16000:] = True
    f, taps = O.beats_forward(x, sd, cfg, padding_mask=pm)
    assert rel_l2(f[:, ::8], g["mask.features_tok8"]) < 1e-5
    # the reference's hook on post_extract_proj holds the tensor the encoder later zeroes IN PLACE at
    # padded tokens (beats.py:359-361, backbone.py:169-170): the tap is the masked tensor
    frames = 1 + (32000 - 400) // 160
    fpad = O.forward_padding_mask(96, O.forward_padding_mask(frames, pm))
    tap0 = np.where(fpad[..., None], np.float32(0), taps["backbone.post_extract_proj"])
    e = np.concatenate([tap0.mean(1), taps["backbone.encoder.layers.11.fc2"].mean(1)], 1)
    assert rel_l2(e, g["mask.mean"]) < 1e-5


def test_eat_frontend_oracle_known_answers():
    """EAT frontend restatement (eat/audio_processor.py:72-143): shape contract, padded rows, truncation, and the two
    invariances that follow from the kaldi per-frame DC removal (a clip offset and the clip-mean removal change nothing
    beyond fp32 rounding).  torchaudio is absent here, so this function has no golden of its own (see its docstring)."""
    from avex_amd import synth
    x = synth.noise_clips(2, 80000, seed=3)                       # 5 s: 498 frames -> zero-padded to 1024
    y = O.eat_preprocess(x)
    assert y.shape == (2, 1024, 128) and y.dtype == np.float32
    pad = (0.0 + 4.268) / (2 * 4.569)
    assert np.allclose(y[:, 498:], pad, atol=1e-6) and not np.allclose(y[:, 497], pad, atol=1e-3)
    # the valid rows are the pinned fbank with a Hann window, no 2**15 scale
    ref = (O.fbank(x - x.mean(1, keepdims=True, dtype=np.float32), window=O.hann_window(400)) + 4.268) / (2 * 4.569)
    assert np.allclose(y[:, :498], ref, atol=1e-5)
    # the BEATs scale only shifts the log-mel by log((2**15)**2) wherever neither side sits on the fp32-eps floor
    xc = x - x.mean(1, keepdims=True, dtype=np.float32)
    plain, scaled = O.fbank(xc, window=O.hann_window(400)), O.fbank(xc * np.float32(2 ** 15), window=O.hann_window(400))
    above = plain > np.log(O.F32_EPS) + 1.0
    assert above.mean() > 0.9 and np.abs((scaled - 2 * np.log(2.0 ** 15)) - plain)[above].max() < 2e-3
    # a DC offset is removed twice over
    y_dc = O.eat_preprocess(x + np.float32(0.25))
    assert np.abs(y_dc[:, :498] - y[:, :498]).max() < 5e-3
    long = synth.noise_clips(1, 170000, seed=4)                   # 1060 frames -> cut to 1024
    assert O.eat_preprocess(long).shape == (1, 1024, 128)
    z = O.eat_preprocess(x, norm_mean=0.0, norm_std=1.0)          # per-sample statistics branch
    assert abs(float(z[0].mean())) < 1e-5 and abs(float(z[0].std(ddof=1)) - 0.5) < 1e-4



def test_stft_oracle_matches_torch_stft():
    """The STFT stage of the spectrogram frontend is pinned against torch.stft itself (the call the reference makes,
    audio_utils.py:139-148): EfficientNet settings (n_fft 800, hop 160, periodic Hann, centre / reflect) and a hamming,
    uncentred, shorter-window variant.  The MelScale matrix has no such pin (torchaudio absent): shape and partition checks."""
    import torch
    from avex_amd import synth
    x = synth.noise_clips(2, 16000, seed=51)
    for n_fft, hop, win, kind, center in ((800, 160, 800, "hann", True), (512, 128, 400, "hamming", False)):
        tw = torch.hann_window(win) if kind == "hann" else torch.hamming_window(win)
        ref = torch.stft(torch.from_numpy(x), n_fft=n_fft, hop_length=hop, win_length=win, window=tw, center=center, return_complex=True).abs().pow(2).numpy()
        got = O.stft_power(x, n_fft, hop, win, tw.numpy(), center)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-5 * ref.max()
    fb = O.htk_mel_fb(401, 128, 16000)
    assert fb.shape == (401, 128) and (fb >= 0).all() and fb.max() <= 1.0
    assert (fb > 0).sum(0).min() >= 1                       # every triangle reaches at least one bin at n_fft 800
    peaks = fb.argmax(0)
    assert (np.diff(peaks) >= 0).all() and peaks[0] >= 0 and peaks[-1] <= 400
    y = O.audio_processor(x, n_fft=800, hop=160)
    assert y.shape == (2, 128, 101) and abs(float(y[0].min())) < 1e-7 and abs(float(y[0].max()) - 1.0) < 1e-6


@pytest.mark.parametrize("variant", sorted(synth.BEATS_VARIANTS))
def test_config_space_variants(golden_dir, variant):
    """BEATsConfig options no official checkpoint uses (pre-LN blocks, the other FFN activations, the gated linear unit, a
    patch-embedding bias, no / ungated relative position bias, no post_extract_proj): the oracle against the real reference's outputs
    (tests/golden/make_variant_goldens.py) on the small configurations of synth.BEATS_VARIANTS."""
    g = np.load(f"{golden_dir}/variants.npz")
    cfg = synth.BEATS_VARIANTS[variant]
    sd = synth.beats_state_dict(cfg, seed=3)
    x = synth.noise_clips(2, 32000, seed=13)
    f, taps = O.beats_forward(x, sd, cfg)
    assert rel_l2(f[:, ::3], g[f"{variant}.features_tok3"]) < 1e-5
    assert rel_l2(O.pooled(f), g[f"{variant}.pooled"]) < 1e-5
    for i in range(int(cfg["encoder_layers"])):
        t = taps[f"backbone.encoder.layers.{i}.fc2"]
        assert rel_l2(t[:, ::6], g[f"{variant}.fc2.{i}_tok6"]) < 1e-5
        assert rel_l2(t.mean(1), g[f"{variant}.fc2.{i}_mean"]) < 2e-5
    pm = np.zeros((2, 32000), bool); pm[1, 16000:] = True
    fm, _ = O.beats_forward(x, sd, cfg, padding_mask=pm)
    assert rel_l2(fm[:, ::3], g[f"{variant}.features_masked_tok3"]) < 1e-5
