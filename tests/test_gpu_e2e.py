"""End-to-end GPU parity of the BEATs path against the committed reference goldens (`-m gpu`).

Goldens (tests/golden/base_api.npz) were produced by the real reference on the synthetic
checkpoint of avex_amd.synth; the bar is BASELINE.json's: pooled 768-d embedding within 1e-3
relative (||a-b||2/||b||2) of the reference's fp32 CPU path.
"""
import os

import numpy as np
import pytest
import torch

from _util import rel_l2
from avex_amd import synth
from oracle import beats_oracle as O

pytestmark = pytest.mark.gpu

POOLED_TOL = {"f16": 1e-3, "bf16": 6e-3}   # bf16 operands: weight rounding alone costs ~2e-3


@pytest.fixture(scope="module")
def base_sd():
    return synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)


@pytest.fixture(scope="module", params=["f16", "bf16", "f16-halfres", "bf16-halfres", "f16-halfres-nofold", "bf16-halfres-nofold",
                                        "f16-halfres-plainq", "bf16-plainq", "f16-halfres-alwaysfold", "bf16-halfres-alwaysfold"])
def encoder(request, built_lib, base_sd):
    """f32 residual stream (generic GEMM epilogues), half residual stream (default: the LayerNorms folded into the streaming GEMM's
    epilogues for chunks of >= 4096 token rows: the cases below run unfolded, the full-batch golden test folded), half residual stream
    with LayerNorm kernels (AVEX_AMD_LN_FOLD=0) or with the fold at every size (=1), and the attention fed with plain Q instead of
    log2(e) Q (AVEX_AMD_Q_LOG2E=0); the knobs are read when the handle is created."""
    import os
    from avex_amd import kernels as K
    dt = request.param.split("-")[0]
    knobs = {"AVEX_AMD_LN_FOLD": "0" if request.param.endswith("nofold") else ("1" if request.param.endswith("alwaysfold") else None),
             "AVEX_AMD_Q_LOG2E": "0" if request.param.endswith("plainq") else None}
    old = {k: os.environ.get(k) for k in knobs}
    for k, v in knobs.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    try:
        enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype=dt, max_chunk_clips=3,
                             residual="half" if "halfres" in request.param else "f32")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    enc.dtype_name = dt
    yield enc
    enc.close()


@pytest.mark.parametrize("tag,B,T", [("b1", 1, 160000), ("b4", 4, 160000), ("odd", 2, 123457), ("short", 3, 16000)])
def test_pooled_and_hooks_match_reference(encoder, golden_dir, tag, B, T):
    g = np.load(f"{golden_dir}/base_api.npz")
    wav = torch.from_numpy(synth.noise_clips(B, T, seed=0)).cuda()
    r = encoder.forward(wav, hook_layers=range(13), want_features=True, want_pooled=True)
    tol = POOLED_TOL[encoder.dtype_name]
    pooled = r["pooled"].cpu().numpy()
    feats = r["features"].cpu().numpy()
    for b in range(B):
        assert rel_l2(pooled[b], g[f"{tag}.pooled"][b]) < tol
    assert rel_l2(feats.mean(1), g[f"{tag}.pooled"]) < tol
    assert rel_l2(feats[:, ::16], g[f"{tag}.feat_tok16"]) < 4 * tol          # frame level (no token averaging)
    all_mean = np.concatenate([r["hooks"][i].cpu().numpy().mean(1) for i in range(13)], axis=1)
    assert rel_l2(all_mean, g[f"{tag}.all_mean"]) < tol
    # pooled hooks computed on device agree with pooling the full taps
    r2 = encoder.forward(wav, hook_layers=[0, 12], hook_pooled=True, want_features=False, want_pooled=True)
    assert rel_l2(r2["hooks"][0].cpu().numpy(), r["hooks"][0].cpu().numpy().mean(1)) < 1e-5
    assert rel_l2(r2["hooks"][12].cpu().numpy(), r["hooks"][12].cpu().numpy().mean(1)) < 1e-5
    assert rel_l2(r2["pooled"].cpu().numpy(), pooled) < 1e-6


def test_pooled_tap_aggregations(encoder):
    """hook_pooled = "mean" / "max" / "cls_token": taps reduced over the tokens on the device (inside the producing GEMM's epilogue for
    clips of >= 64 tokens, through a scratch tap for shorter ones) equal the same reduction of the full taps -- max and first token bit
    for bit.  Hook 0 (post_extract_proj) takes the scratch route at every length."""
    for samples in (160000, 16000):          # 496 tokens (fused), 48 tokens (scratch)
        wav = torch.from_numpy(synth.noise_clips(3, samples, seed=23)).cuda()
        full = encoder.forward(wav, hook_layers=[0, 1, 7, 12], want_features=False)["hooks"]
        for agg in ("mean", "max", "cls_token"):
            r = encoder.forward(wav, hook_layers=[0, 1, 7, 12], hook_pooled=agg, want_features=False)["hooks"]
            for i in (0, 1, 7, 12):
                want = {"mean": full[i].mean(1), "max": full[i].max(1)[0], "cls_token": full[i][:, 0]}[agg]
                assert r[i].shape == want.shape
                if agg == "mean":
                    assert rel_l2(r[i].cpu().numpy(), want.cpu().numpy()) < 1e-5
                else:
                    assert torch.equal(r[i], want), (samples, agg, i)
        # "none" is a non-empty string and code 0: the taps come back FULL, in buffers sized for full taps (sized by the string's
        # truthiness they were [B, E] buffers the library wrote B * T' * E floats into)
        r = encoder.forward(wav, hook_layers=[0, 12], hook_pooled="none", want_features=False)["hooks"]
        for i in (0, 12):
            assert r[i].shape == full[i].shape and torch.equal(r[i], full[i])


def test_layernorm_in_split_k_epilogue(encoder, monkeypatch):
    """Below the fold threshold the two LayerNorms of a layer ride in the split-K epilogue of out_proj / fc2 (GemmArgs::post_ln_*):
    against the same forward with LayerNorm kernels (AVEX_AMD_POST_LN=0) -- the same arithmetic on the same rounded sums, another order
    of the row reductions -- for one clip, three clips, a short clip, hooks and features."""
    for B, samples in ((1, 160000), (3, 48000), (2, 8000)):
        wav = torch.from_numpy(synth.noise_clips(B, samples, seed=31)).cuda()
        a = encoder.forward(wav, hook_layers=[0, 6, 12], want_features=True, want_pooled=True)
        monkeypatch.setenv("AVEX_AMD_POST_LN", "0")
        b = encoder.forward(wav, hook_layers=[0, 6, 12], want_features=True, want_pooled=True)
        monkeypatch.delenv("AVEX_AMD_POST_LN")
        tol = 2e-4 if encoder.dtype_name == "f16" else 1.5e-3
        assert rel_l2(a["pooled"].cpu().numpy(), b["pooled"].cpu().numpy()) < tol
        assert rel_l2(a["features"].cpu().numpy(), b["features"].cpu().numpy()) < 4 * tol
        for i in (0, 6, 12):
            assert rel_l2(a["hooks"][i].cpu().numpy(), b["hooks"][i].cpu().numpy()) < 4 * tol


def test_chunking_is_invisible(encoder):
    """max_chunk_clips=3 above: a batch of 7 runs as 3+3+1 and must equal per-clip runs."""
    wav = torch.from_numpy(synth.noise_clips(7, 32000, seed=11)).cuda()
    full = encoder.forward(wav, want_pooled=True)["pooled"].cpu().numpy()
    for b in (0, 3, 6):
        one = encoder.forward(wav[b:b + 1], want_pooled=True)["pooled"].cpu().numpy()
        assert np.array_equal(one[0], full[b])       # clips are independent: bit-identical


def test_padding_mask_matches_reference(encoder, golden_dir):
    g = np.load(f"{golden_dir}/base_api.npz")
    x = synth.noise_clips(2, 32000, seed=5)
    pm = np.zeros((2, 32000), bool); pm[1, 16000:] = True
    frames = 1 + (32000 - 400) // 160
    fpad = O.forward_padding_mask(96, O.forward_padding_mask(frames, pm))   # beats.py:346-347,355-356
    r = encoder.forward(torch.from_numpy(x).cuda(), hook_layers=[0, 12], want_features=True,
                        frame_pad=torch.from_numpy(fpad))
    tol = POOLED_TOL[encoder.dtype_name]
    assert rel_l2(r["features"].cpu().numpy()[:, ::8], g["mask.features_tok8"]) < 4 * tol
    emb = np.concatenate([r["hooks"][0].cpu().numpy().mean(1), r["hooks"][12].cpu().numpy().mean(1)], axis=1)
    assert rel_l2(emb, g["mask.mean"]) < tol


def test_oracle_agrees_on_fresh_input(encoder, base_sd):
    """A case no golden covers: GPU path vs the CPU oracle on a new seed and length."""
    x = synth.noise_clips(2, 48000, seed=21)
    f, taps = O.beats_forward(x, base_sd, synth.BEATS_BASE_CFG)
    r = encoder.forward(torch.from_numpy(x).cuda(), hook_layers=[5], want_pooled=True)
    tol = POOLED_TOL[encoder.dtype_name]
    assert rel_l2(r["pooled"].cpu().numpy(), O.pooled(f)) < tol
    assert rel_l2(r["hooks"][5].cpu().numpy().mean(1), taps["backbone.encoder.layers.4.fc2"].mean(1)) < tol


def test_nan_input_stays_nan_and_stays_in_its_clip(encoder):
    """A NaN sample must not be laundered into a finite embedding: the reference (fp32 torch) returns NaN for such a clip -- features, pooled
    vector and every hooked fc2 output, which is what extract_embeddings hands back (backbone.py:350-375) -- and so must this path, with the
    other clips of the batch bit for bit what they are without it.  Round 5 found four places that turned the NaN into ordinary numbers: the
    software f16 clamp and the filterbank's log floor (fmin / fmax return their non-NaN operand), the MFMAs under MODE.FP16_OVFL (a NaN
    operand counts as 0: scripts/micro/mfma_nan.hip) and the half-output GELU's clamps (csrc/common.h)."""
    x = synth.noise_clips(4, 32000, seed=31)
    layers = [0, 1, 6, 12]
    clean = encoder.forward(torch.from_numpy(x).cuda(), hook_layers=layers, want_features=True, want_pooled=True)
    cp, cf = clean["pooled"].cpu().numpy(), clean["features"].cpu().numpy()
    ch = {i: clean["hooks"][i].cpu().numpy() for i in layers}
    bad = x.copy()
    bad[2, 17000] = np.nan
    r = encoder.forward(torch.from_numpy(bad).cuda(), hook_layers=layers, want_features=True, want_pooled=True)
    p, f = r["pooled"].cpu().numpy(), r["features"].cpu().numpy()
    assert np.isnan(p[2]).all() and np.isnan(f[2]).any()
    for i in layers:
        h = r["hooks"][i].cpu().numpy()
        assert np.isnan(h[2]).any(), f"hook {i} of the poisoned clip is finite"
        if i >= 1:      # behind the first attention every token of the clip has met the NaN key
            assert np.isnan(h[2]).all(), f"hook {i}: {int(np.isfinite(h[2]).sum())} finite values"
        for c in (0, 1, 3):
            assert np.array_equal(h[c], ch[i][c])
    for i in (0, 1, 3):
        assert np.array_equal(p[i], cp[i]) and np.array_equal(f[i], cf[i])


def test_degenerate_signals(encoder, base_sd):
    """Signals at the ends of the frontend's range, GPU path vs the CPU oracle: digital silence (every mel bin at the log floor,
    every LayerNorm row of the patch embedding constant), a DC offset (removed per frame), one impulse, a full-scale square wave,
    white noise at the reference tests' own amplitude (std 1, i.e. 8x beyond full scale: tests/unittests/test_batched_fbank.py:62-64)
    and a clip that is silent in its second half.  Nothing may overflow the 16-bit operands or turn into NaN."""
    n = 32000
    rng = np.random.default_rng(17)
    x = np.zeros((6, n), np.float32)
    x[1] = 0.25
    x[2, 12345] = 1.0
    x[3] = np.where((np.arange(n) // 40) % 2 == 0, 1.0, -1.0)
    x[4] = rng.standard_normal(n).astype(np.float32)
    x[5, :n // 2] = 0.1 * rng.standard_normal(n // 2).astype(np.float32)
    f, _ = O.beats_forward(x, base_sd, synth.BEATS_BASE_CFG)
    r = encoder.forward(torch.from_numpy(x).cuda(), want_features=True, want_pooled=True)
    got = r["pooled"].cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(r["features"].cpu().numpy()).all()
    want = O.pooled(f)
    # bf16 operands: these signals sit at the edge of the bf16 bar (the half-silent clip measures 5.5e-3 .. 6.2e-3 depending on the
    # residual mode and on where Q is rounded: its silent half is LayerNorm of almost constant rows, which amplifies operand rounding);
    # the f16 bar -- the one north_star states -- is not loosened
    tol = POOLED_TOL[encoder.dtype_name] * (1.25 if encoder.dtype_name == "bf16" else 1.0)
    for i in range(x.shape[0]):
        assert rel_l2(got[i:i + 1], want[i:i + 1]) < tol, (i, rel_l2(got[i:i + 1], want[i:i + 1]))


def test_edge_sizes(encoder, base_sd):
    """Smallest and unusual inputs the path takes, and the ones it refuses loudly: 16 frames -> 8 tokens, 1024 frames -> 512
    tokens (the last size of the single-block attention path), one token row more (520 tokens: two query blocks, three key
    blocks), fewer samples than one analysis window, and an empty batch."""
    from avex_amd._capi import AvexHipError
    tol = POOLED_TOL[encoder.dtype_name]
    for samples, tokens in ((400 + 160 * 15, 8), (400 + 160 * 1023, 512), (400 + 160 * 1039, 520)):
        x = synth.noise_clips(1, samples, seed=31)
        f, _ = O.beats_forward(x, base_sd, synth.BEATS_BASE_CFG)
        assert f.shape[1] == tokens
        r = encoder.forward(torch.from_numpy(x).cuda(), want_features=True, want_pooled=True)
        assert r["features"].shape == (1, tokens, 768)
        assert rel_l2(r["pooled"].cpu().numpy(), O.pooled(f)) < (3 * tol if tokens == 8 else tol)   # 8 tokens: no averaging
    with pytest.raises(AvexHipError):
        encoder.forward(torch.zeros(2, 300, device="cuda"), want_pooled=True)                       # shorter than one window
    with pytest.raises((AvexHipError, ValueError)):
        encoder.forward(torch.zeros(0, 160000, device="cuda"), want_pooled=True)                    # empty batch


def test_long_clips_match_oracle(built_lib, base_sd):
    """Clips longer than 10.3 s (the reference takes any length, beats.py:344-361 / backbone.py:151-221): 20 s = 992 tokens and
    a 12.1 s clip beside a padding mask, BEATs-base, against the CPU oracle; chunking by token count must not show."""
    from avex_amd import kernels as K
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype="f16", max_chunk_clips=2)
    x = synth.noise_clips(3, 320000, seed=41)
    f, taps = O.beats_forward(x[:1], base_sd, synth.BEATS_BASE_CFG)
    r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[0, 12], want_features=True, want_pooled=True)
    assert r["features"].shape == (3, 992, 768)
    assert rel_l2(r["pooled"][:1].cpu().numpy(), O.pooled(f)) < 1e-3
    assert rel_l2(r["features"][:1].cpu().numpy(), f) < 4e-3
    assert rel_l2(r["hooks"][12][:1].cpu().numpy().mean(1), taps["backbone.encoder.layers.11.fc2"].mean(1)) < 1e-3
    one = enc.forward(torch.from_numpy(x[2:3]).cuda(), want_pooled=True)["pooled"]
    assert torch.equal(one[0], r["pooled"][2])                              # chunks of 1 clip (2 * 512 / 992) vs alone
    # 12.1 s with the last third padded
    T = 400 + 160 * 1209
    x2 = synth.noise_clips(1, T, seed=42)
    pm = np.zeros((1, T), bool); pm[0, 2 * T // 3:] = True
    f2, _ = O.beats_forward(x2, base_sd, synth.BEATS_BASE_CFG, padding_mask=pm)
    nt = f2.shape[1]
    fpad = O.forward_padding_mask(nt, O.forward_padding_mask(1210, pm))
    r2 = enc.forward(torch.from_numpy(x2).cuda(), want_features=True, frame_pad=torch.from_numpy(fpad))
    assert r2["features"].shape == (1, nt, 768) and nt == 600
    assert rel_l2(r2["features"].cpu().numpy(), f2) < 4e-3
    enc.close()


def test_full_size_properties(built_lib, base_sd):
    """BASELINE config C2 shape (batch 256 x 10 s) is too big for the CPU oracle; check size-independent
    properties instead: clip independence / permutation equivariance and agreement with small-batch runs."""
    from avex_amd import kernels as K
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype="f16")
    B = 64
    wav = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
    p = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    p2 = enc.forward(wav[perm], want_features=False, want_pooled=True)["pooled"]
    assert torch.equal(p2, p[perm])
    assert torch.isfinite(p).all()
    # pooled-only batches of >= 32 clips take the fused final LayerNorm + mean kernel; it must agree with pooling the fp32 features
    sub = wav[:40]
    f = enc.forward(sub, want_features=True, want_pooled=True)
    assert rel_l2(p[:40].cpu().numpy(), f["features"].mean(1).cpu().numpy()) < 2e-6
    assert rel_l2(p[:40].cpu().numpy(), f["pooled"].cpu().numpy()) < 2e-6
    enc.close()


def test_model_without_post_extract_proj(built_lib):
    """embed_dim == encoder_embed_dim: no post_extract_proj (beats.py:357-358).  The padded tokens must still be zeroed before the
    pos-conv (backbone.py:169-170), hook 0 does not exist and asking for it is refused, not silently left unwritten."""
    from avex_amd import kernels as K
    from avex_amd._capi import AvexHipError
    cfg = dict(synth.BEATS_BASE_CFG, embed_dim=768, encoder_layers=2)
    sd = synth.beats_state_dict(cfg, seed=4)
    assert "backbone.post_extract_proj.weight" not in sd
    x = synth.noise_clips(2, 32000, seed=6)
    pm = np.zeros((2, 32000), bool); pm[1, 20000:] = True
    f_ref, taps = O.beats_forward(x, sd, cfg, padding_mask=pm)
    frames = 1 + (32000 - 400) // 160
    fpad = O.forward_padding_mask(96, O.forward_padding_mask(frames, pm))
    for residual in ("half", "f32"):
        enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", residual=residual)
        r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[2], want_features=True, frame_pad=torch.from_numpy(fpad))
        assert rel_l2(r["features"].cpu().numpy(), f_ref) < 3e-3
        assert rel_l2(r["hooks"][2].cpu().numpy().mean(1), taps["backbone.encoder.layers.1.fc2"].mean(1)) < 1e-3
        with pytest.raises(AvexHipError):
            enc.forward(torch.from_numpy(x).cuda(), hook_layers=[0], want_features=True)
        enc.close()


def test_config_c2_full_size_vs_reference_golden(built_lib, base_sd, golden_dir, monkeypatch):
    """BASELINE config C2 at its real size and in the bench's exact configuration (f16 operands, operand-type residual stream,
    one 256-clip chunk, pooled-only -> the fused LayerNorm + mean kernel, M = 126 976 rows through the persistent GEMM): the four
    clips of the reference golden b4 sit at rows 0, 1, 128 and 255 of the batch and must come out within north_star's 1e-3 of
    the reference's fp32 CPU result; where a clip sits in the batch, and how large the batch is, must not matter."""
    from avex_amd import kernels as K
    g = np.load(f"{golden_dir}/base_api.npz")["b4.pooled"]
    gold = synth.noise_clips(4, 160000, seed=0)
    rows = (0, 1, 128, 255)
    x = synth.noise_clips(256, 160000, seed=0, first_clip=1000)
    for k, r in enumerate(rows):
        x[r] = gold[k]
    wav = torch.from_numpy(x).cuda()
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype="f16", max_chunk_clips=256, residual="half")
    p = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    assert p.shape == (256, 768) and torch.isfinite(p).all()
    got = p[list(rows)].cpu().numpy()
    assert rel_l2(got, g) < 1e-3                                           # all four together
    for k in range(4):
        assert rel_l2(got[k], g[k]) < 1e-3                                 # and each clip alone
    # the same clips at other rows of the same-size batch: bit-identical
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(5)).cuda()
    p2 = enc.forward(wav[perm], want_features=False, want_pooled=True)["pooled"]
    assert torch.equal(p2, p[perm])
    # the same clips as a batch of 4: below 4 096 token rows the default policy runs LayerNorm kernels and the 128-tile GEMM instead of the
    # folded streaming kernel -- other roundings of the same arithmetic, both inside the bar against the reference
    small = enc.forward(torch.from_numpy(gold).cuda(), want_features=True, want_pooled=True)
    assert rel_l2(small["pooled"].cpu().numpy(), g) < 1e-3
    assert rel_l2(got, small["pooled"].cpu().numpy()) < 2.2e-4          # measured 1.08e-4 (profiles/r04_parity.json "batch_dependence"); bit-identical with batch_invariant=True
    assert rel_l2(small["pooled"].cpu().numpy(), small["features"].mean(1).cpu().numpy()) < 2e-6
    enc.close()
    # AVEX_AMD_LN_FOLD=1 (fold at every size): the batch of 4 then takes the same kernels as the batch of 256 and differs only by the
    # unfused final LayerNorm + pooling of fp32 features
    monkeypatch.setenv("AVEX_AMD_LN_FOLD", "1")
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype="f16", max_chunk_clips=256, residual="half")
    big = enc.forward(wav, want_features=False, want_pooled=True)["pooled"][list(rows)].cpu().numpy()
    small = enc.forward(torch.from_numpy(gold).cuda(), want_features=True, want_pooled=True)
    assert rel_l2(big, g) < 1e-3
    assert rel_l2(big, small["pooled"].cpu().numpy()) < 5e-5
    assert rel_l2(big, small["features"].mean(1).cpu().numpy()) < 5e-5
    enc.close()


@pytest.mark.parametrize("residual", ["half", "f32"])
def test_batch_invariant_mode_same_bits_alone_and_in_256(built_lib, base_sd, residual):
    """batch_invariant=True (kernels.residual_code: LayerNorm fold at every size, no split-K, one final LayerNorm + pool path): a clip's
    pooled embedding and its mean-pooled taps are bit-identical whether it comes alone, in a batch of 4 or at any row of a batch of 256
    -- what the reference's fp32 path gives by construction (beats_model.py:279-429).  The default mode trades that for quicker
    small-batch kernels (measured difference in profiles/r04_parity.json, bounded in test_config_c2_full_size_vs_reference_golden)."""
    from avex_amd import kernels as K
    x = torch.from_numpy(synth.noise_clips(256, 160000, seed=0)).cuda()
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd, operand_dtype="f16", residual=residual, batch_invariant=True)
    big = enc.forward(x, hook_layers=[0, 6, 12], hook_pooled="mean", want_features=False, want_pooled=True)
    for rows in ([17], [0, 1, 128, 255]):
        small = enc.forward(x[rows], hook_layers=[0, 6, 12], hook_pooled="mean", want_features=False, want_pooled=True)
        assert torch.equal(small["pooled"], big["pooled"][rows]), (residual, rows)
        for i in (0, 6, 12):
            assert torch.equal(small["hooks"][i], big["hooks"][i][rows]), (residual, rows, i)
    # frames too: a clip's features alone and in a batch of 8
    f8 = enc.forward(x[:8], want_features=True)["features"]
    f1 = enc.forward(x[3:4], want_features=True)["features"]
    assert torch.equal(f1[0], f8[3])
    enc.close()


def test_config_c5_full_size_efficientnet(built_lib):
    """BASELINE config C5 at its real size (1024 clips x 10 s through the mel frontend and EfficientNet-B0): clips are independent,
    so two clips run alone must reproduce their rows of the 1024-clip batch."""
    from avex_amd import kernels as K
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    enc = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
    plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
    wav = torch.from_numpy(synth.noise_clips(1024, 160000, seed=2)).cuda()
    full = enc.forward(plan(wav), want_features=False, want_pooled=True)["pooled"]
    assert full.shape == (1024, 1280) and torch.isfinite(full).all()
    for r in (0, 517, 1023):
        one = enc.forward(plan(wav[r:r + 1]), want_features=False, want_pooled=True)["pooled"]
        # bit-identical since the squeeze-excitation pool became ordered per-workgroup partial sums (round 3; it was fp32 atomics before,
        # 4e-5 apart from run to run): every kernel of this family works clip by clip in a fixed order (profiles/r04_parity.json)
        assert torch.equal(one, full[r:r + 1])
    # ... and at full size against the ORACLE, not only against itself: the two clips of tests/golden/family_small.npz (oracle/effnet_oracle.py on the
    # synthetic checkpoint; UNPINNED: torchvision is absent from the reference tree) at rows 0 and 1023 of the 1024-clip batch
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "family_small.npz"))
    g = torch.from_numpy(synth.noise_clips(2, int(gold["effnet.samples"][0]), seed=int(gold["effnet.seed"][0]))).cuda()
    wav[0] = g[0]; wav[1023] = g[1]
    full = enc.forward(plan(wav), want_features=False, want_pooled=True)["pooled"]
    got = torch.stack([full[0], full[1023]]).cpu().numpy()
    assert rel_l2(got, gold["effnet.pooled"]) < 1e-3


def test_config_c5_nan_stays_in_its_clip(built_lib):
    """EfficientNet path (mel frontend -> stem -> fused block fronts -> GEMMs -> pool): a NaN sample makes ITS clip's pooled vector NaN -- the
    reference's torchvision stack in fp32 propagates it (efficientnet.py:163-215) -- and leaves the other clips bit for bit unchanged.  The fused
    block front dropped it in round 4: its MFMAs ran with MODE.FP16_OVFL set, which makes them read a NaN operand as 0 (scripts/micro/mfma_nan.hip)."""
    from avex_amd import kernels as K
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    enc = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
    plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
    x = synth.noise_clips(4, 160000, seed=5)
    clean = enc.forward(plan(torch.from_numpy(x).cuda()), want_features=False, want_pooled=True)["pooled"].cpu().numpy()
    bad = x.copy()
    bad[2, 80000] = np.nan
    p = enc.forward(plan(torch.from_numpy(bad).cuda()), want_features=False, want_pooled=True)["pooled"].cpu().numpy()
    assert np.isnan(p[2]).all()
    for i in (0, 1, 3):
        assert np.array_equal(p[i], clean[i])


def test_bias_table_cache_is_bounded_and_eviction_is_invisible(built_lib, base_sd):
    """Variable-length inference: each token count needs its own Toeplitz bias table (api.cpp bias_tab_for); the cache holds 16 of them,
    least recently used out.  40 different lengths, then the first one again: same bits as its first run."""
    from avex_amd import kernels as K
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, base_sd)
    x0 = torch.from_numpy(synth.noise_clips(1, 16000, seed=21)).cuda()
    first = enc.forward(x0, want_features=False, want_pooled=True)["pooled"].clone()
    for i in range(40):
        T = 16000 + 2560 * (i + 1)                                   # 16 more frames = 8 more tokens each
        r = enc.forward(torch.from_numpy(synth.noise_clips(1, T, seed=22 + i)).cuda(), want_features=False, want_pooled=True)
        assert bool(torch.isfinite(r["pooled"]).all())
    again = enc.forward(x0, want_features=False, want_pooled=True)["pooled"]
    assert torch.equal(first, again)
    enc.close()


@pytest.mark.parametrize("mode", ["f16-f32", "f16-half", "bf16-half"])
@pytest.mark.parametrize("variant", sorted(synth.BEATS_VARIANTS))
def test_config_space_variants(built_lib, golden_dir, variant, mode):
    """BEATsConfig options no official checkpoint uses (beats.py:181-196): pre-LN blocks with the LayerNorm after the stack, ReLU / tanh-form
    GELU / tanh / linear FFNs, the gated linear unit, a patch-embedding bias, no / ungated relative position bias, no post_extract_proj.
    The HIP handle against the REAL reference's outputs (tests/golden/make_variant_goldens.py; the oracle is pinned to the same file in
    tests/test_oracle_golden.py), features, pooled, every hook tap and the half-padded batch."""
    from avex_amd import kernels as K
    dt, res = mode.split("-")
    g = np.load(f"{golden_dir}/variants.npz")
    cfg = synth.BEATS_VARIANTS[variant]
    sd = synth.beats_state_dict(cfg, seed=3)
    L = int(cfg["encoder_layers"])
    has_post = int(cfg["embed_dim"]) != int(cfg["encoder_embed_dim"])
    enc = K.BeatsEncoder(cfg, sd, operand_dtype=dt, residual=res)
    try:
        x = torch.from_numpy(synth.noise_clips(2, 32000, seed=13)).cuda()
        hooks = list(range(1, L + 1))
        r = enc.forward(x, hook_layers=hooks, want_features=True, want_pooled=True)
        tol = POOLED_TOL[dt]
        feats = r["features"].cpu().numpy()
        assert feats.shape == (2, 96, int(cfg["encoder_embed_dim"]))
        assert rel_l2(r["pooled"].cpu().numpy(), g[f"{variant}.pooled"]) < tol
        assert rel_l2(feats[:, ::3], g[f"{variant}.features_tok3"]) < 4 * tol
        for i in range(L):
            tap = r["hooks"][i + 1].cpu().numpy()
            assert rel_l2(tap.mean(1), g[f"{variant}.fc2.{i}_mean"]) < 2 * tol
            assert rel_l2(tap[:, ::6], g[f"{variant}.fc2.{i}_tok6"]) < 4 * tol
        if not has_post:
            with pytest.raises(K.AvexHipError):
                enc.forward(x, hook_layers=[0], want_pooled=True)
        pm = np.zeros((2, 32000), bool); pm[1, 16000:] = True
        frames = 1 + (32000 - 400) // 160
        fpad = O.forward_padding_mask(96, O.forward_padding_mask(frames, pm))
        rm = enc.forward(x, want_features=True, frame_pad=torch.from_numpy(fpad))
        assert rel_l2(rm["features"].cpu().numpy()[:, ::3], g[f"{variant}.features_masked_tok3"]) < 4 * tol
        # pooled-only path at a batch the fused LayerNorm + mean kernel takes (>= 32 clips): every clip equals its single-clip run
        xb = x[:1].repeat(33, 1)
        pb = enc.forward(xb, want_pooled=True)["pooled"].cpu().numpy()
        assert rel_l2(pb[-1:], g[f"{variant}.pooled"][:1]) < tol
    finally:
        enc.close()
