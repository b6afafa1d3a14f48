"""EAT on the GPU (BASELINE config C3): kernels, encoder and model class against oracle/eat_oracle.py
(parity unpinned vs the HF remote model, see that file)."""
import numpy as np
import pytest
import torch

import avex_amd
from avex_amd import synth
from avex_amd import kernels as K
from oracle import beats_oracle as O
from oracle import eat_oracle as EO
from tests._util import rel_l2

pytestmark = pytest.mark.gpu


def test_token_embed_ln_matches_oracle(built_lib):
    B, n, E = 3, 512, 768
    pe = synth.normal("tepe", (B * n, E), 1.0)
    pos = synth.sincos_2d_positions(E, 64, 8)
    cls = synth.normal("tecls", (E,), 0.5); w = 1.0 + synth.normal("tew", (E,), 0.1); b = synth.normal("teb", (E,), 0.1)
    peh = torch.from_numpy(pe).cuda().half()
    outh, outf = K.token_embed_ln(peh, torch.from_numpy(pos).cuda(), torch.from_numpy(cls).cuda(), torch.from_numpy(w).cuda(),
                                  torch.from_numpy(b).cuda(), 1e-6, B, want_f32=True)
    x = peh.float().cpu().numpy().reshape(B, n, E) + pos[None]
    x = np.concatenate([np.broadcast_to(cls, (B, 1, E)), x], 1)
    ref = O.layer_norm(x.astype(np.float32), w, b, 1e-6).reshape(B * (n + 1), E)
    assert rel_l2(outf.cpu().numpy(), ref) < 5e-6
    assert rel_l2(outh.float().cpu().numpy(), ref) < 6e-4


def test_fbank_patches_match_frontend(built_lib):
    """The filterbank kernel's patch-major output = EATAudioProcessor's image cut into 16 x 16 patches (5 s clips: 498 frames,
    the rest of the 1024 rows is the normalised zero padding)."""
    wav = synth.noise_clips(3, 80000, seed=8)
    ref = EO.patchify(O.eat_preprocess(wav), 16)                                   # [3, 512, 256]
    plan = K.FbankPlan(win_length=400, hop_length=160, n_mels=128, input_scale=1.0, norm_mean=-4.268, norm_div=2 * 4.569,
                       window=K.hann_window(400), mel_fb=K.kaldi_mel_filterbank(128, 512, 16000.0, 20.0, 0.0))
    got = plan.patches(torch.from_numpy(wav).cuda(), out_frames=1024, patch=16, remove_clip_mean=True).float().cpu().numpy()
    assert got.shape == (3 * 512, 256)
    assert np.abs(got.reshape(3, 512, 256) - ref).max() < 2e-3                     # f16 rounding of values in [-2.5, 2.5]


@pytest.mark.parametrize("dtype,tol,ftol", [("f16", 1.0e-3, 1.0e-3), ("bf16", 6.0e-3, 1.5e-2)])
def test_eat_encoder_matches_oracle(built_lib, dtype, tol, ftol):
    """EAT-base, all 12 blocks, 513 tokens (two query blocks / three key blocks in the attention), 2 clips of 5 s.
    The pooled bar is north_star's 1e-3 (f16); ``ftol`` is the bar of the un-averaged rows (class token -- the reference wrapper's default
    pooling, eat_hf.py:149,281-282 -- and frame level): 1e-3 as well since round 6, when ``residual="auto"`` (BEATs' policy) put every call
    that returns such rows on the fp32 residual stream (round 5 on the operand-type stream: class token 1.12e-3, frame level 1.17e-3;
    profiles/r06_parity.json has both)."""
    from avex_amd.eat_encoder import EatEncoder
    cfg = synth.EAT_BASE_CFG
    sd = synth.eat_state_dict(cfg)
    wav = synth.noise_clips(2, 80000, seed=12)
    ref, taps = EO.eat_forward(wav, sd, cfg)
    enc = EatEncoder(cfg, sd, operand_dtype=dtype)
    r = enc.forward(torch.from_numpy(wav).cuda(), hook_layers=[0, 5, 11], pooling="mean")
    f = r["features"].cpu().numpy()
    assert f.shape == ref.shape == (2, 513, 768)
    assert rel_l2(f.mean(1), ref.mean(1)) < tol                                    # pooled embedding
    assert rel_l2(f[:, 0], ref[:, 0]) < ftol                                       # CLS token (no averaging)
    assert rel_l2(f, ref) < ftol
    for i in (0, 5, 11):
        t = taps[f"backbone.model.blocks.{i}.attn.proj"]
        assert rel_l2(r["hooks"][i].cpu().numpy().mean(1), t.mean(1)) < 2 * tol
    assert rel_l2(r["pooled"].cpu().numpy(), ref.mean(1)) < tol
    # token means only (config C3's timed call): the operand-type stream, a second handle of the same encoder
    assert set(enc._handles) == {"f32"}
    po = enc.forward(torch.from_numpy(wav).cuda(), want_features=False, pooling="mean")["pooled"].cpu().numpy()
    assert set(enc._handles) == {"f32", "half"}
    assert rel_l2(po, ref.mean(1)) < tol
    pc = enc.forward(torch.from_numpy(wav).cuda(), want_features=False, pooling="cls")["pooled"].cpu().numpy()      # the class token alone: fp32 stream again
    assert np.array_equal(pc, f[:, 0])
    # the same image handed over as a spectrogram gives the same answer as the fused frontend (f16 patches either way)
    spec = torch.from_numpy(O.eat_preprocess(wav)).cuda()
    r2 = enc.forward(spec=spec, pooling="cls")
    assert rel_l2(r2["features"].cpu().numpy(), f) < max(2e-3, ftol)        # two roundings of the same image to the operand type
    assert torch.equal(r2["pooled"], r2["features"][:, 0])


def test_eat_model_on_gpu(built_lib, tmp_path):
    """The model class through load_model with a local checkpoint: features, hooks + extract_embeddings (every aggregation, the extra
    `pooling` keyword), classifier mode."""
    from safetensors.numpy import save_file
    cfg = dict(synth.EAT_BASE_CFG, depth=2)
    sd = synth.eat_state_dict(cfg)
    x = synth.noise_clips(2, 48000, seed=14)
    ref, taps = EO.eat_forward(x, sd, cfg)
    from avex_amd.eat_hf import EATHFModel
    m = EATHFModel(device="cuda", return_features_only=True, init_config={"depth": 2}).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    f = m(torch.from_numpy(x))
    assert f.shape == (2, 513, 768) and f.is_cuda
    assert rel_l2(f.mean(1).cpu().numpy(), ref.mean(1)) < 1.5e-3
    names = m.register_hooks_for_layers(["all"])
    assert names == [f"backbone.model.blocks.{i}.attn.proj" for i in range(2)]
    t0, t1 = taps[names[0]], taps[names[1]]
    e = m.extract_embeddings({"raw_wav": torch.from_numpy(x)}, aggregation="mean")
    assert e.shape == (2, 1536) and rel_l2(e.cpu().numpy(), np.concatenate([t0.mean(1), t1.mean(1)], 1)) < 2e-3
    e = m.extract_embeddings(torch.from_numpy(x), aggregation="cls_token", pooling="mean")
    assert rel_l2(e.cpu().numpy(), np.concatenate([t0[:, 0], t1[:, 0]], 1)) < 6e-3
    e = m.extract_embeddings(torch.from_numpy(x), aggregation="none")
    assert isinstance(e, list) and len(e) == 2 and e[1].shape == (2, 513, 768)
    assert not m._hook_outputs
    m.deregister_all_hooks()
    # classifier mode: CLS pooling + Linear
    c = EATHFModel(device="cuda", num_classes=5, init_config={"depth": 2}).eval()
    head_w = synth.normal("eathead", (5, 768), 0.05); head_b = synth.normal("eatheadb", (5,), 0.05)
    full = dict(sd, **{"classifier.weight": head_w, "classifier.bias": head_b})
    c.load_state_dict({k: torch.from_numpy(v) for k, v in full.items()})
    logits = c(torch.from_numpy(x))
    assert logits.shape == (2, 5)
    assert rel_l2(logits.cpu().numpy(), ref[:, 0] @ head_w.T + head_b) < 5e-3
    # through the registry with a local safetensors checkpoint
    path = tmp_path / "eat.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in synth.eat_state_dict().items()}, str(path))
    lm = avex_amd.load_model("esp_aves2_eat_all", device="cuda", checkpoint_path=str(path), return_features_only=True)
    out = lm(torch.from_numpy(x))
    assert out.shape == (2, 513, 768) and torch.isfinite(out).all()


def test_config_c3_full_size(built_lib, monkeypatch):
    """BASELINE config C3 at its real size (512 clips x 5 s): finite, and clips are independent -- rows of the big batch equal the
    same clips run in a batch of 3.  Default policy: the small batch runs LayerNorm kernels + the 128-tile GEMM, the big one the folded
    streaming kernel (other roundings of the same arithmetic); with the fold at every size the two take the same kernels."""
    from avex_amd.eat_encoder import EatEncoder
    wav = torch.from_numpy(synth.noise_clips(512, 80000, seed=16)).cuda()
    rows = [0, 300, 511]
    for fold, bar in ((None, 5e-4), ("1", 2e-5)):
        if fold:
            monkeypatch.setenv("AVEX_AMD_LN_FOLD", fold)
        enc = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype="f16")
        full = enc.forward(wav, want_features=False, pooling="mean")["pooled"]
        assert full.shape == (512, 768) and torch.isfinite(full).all()
        small = enc.forward(wav[rows].contiguous(), want_features=False, pooling="mean")["pooled"]
        assert rel_l2(small.cpu().numpy(), full[rows].cpu().numpy()) < bar, fold
        if fold is None:
            # ... and at full size against the ORACLE, not only against itself: the two clips of tests/golden/family_small.npz (oracle/eat_oracle.py on
            # the synthetic checkpoint; UNPINNED: EAT's remote code is absent from the reference tree) at rows 0 and 511 of the 512-clip batch -- token
            # mean on the operand-type stream (the timed path of config C3), class token on the fp32 stream
            import os
            gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "family_small.npz"))
            g = torch.from_numpy(synth.noise_clips(2, int(gold["eat.samples"][0]), seed=int(gold["eat.seed"][0]))).cuda()
            w2 = wav.clone(); w2[0] = g[0]; w2[511] = g[1]
            pm = enc.forward(w2, want_features=False, pooling="mean")["pooled"]
            assert rel_l2(torch.stack([pm[0], pm[511]]).cpu().numpy(), gold["eat.pooled_mean"]) < 1e-3
            pc = enc.forward(w2, want_features=False, pooling="cls")["pooled"]
            assert rel_l2(torch.stack([pc[0], pc[511]]).cpu().numpy(), gold["eat.cls"]) < 1e-3
            del w2
        del enc


def test_nan_sample_stays_nan_and_in_its_clip_eat_and_aves(built_lib):
    """The families that share the BEATs kernels: a NaN sample gives NaN for its clip and leaves the others bit for bit unchanged (EAT: 513 tokens
    through the long-clip attention; AVES: conv feature extractor -- GroupNorm over time -- then the transformer at a length the default attention
    kernel takes without a bias table).  The references compute in fp32 and propagate it (eat_hf.py:241-289, aves_model.py:62-151)."""
    import numpy as np
    from avex_amd.eat_encoder import EatEncoder
    from avex_amd.aves_encoder import AvesEncoder
    enc = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(synth.EAT_BASE_CFG), operand_dtype="f16")
    x = synth.noise_clips(3, 80000, seed=14)
    clean = enc.forward(torch.from_numpy(x).cuda(), hook_layers=[11], pooling="mean")
    bad = x.copy(); bad[1, 40000] = np.nan
    r = enc.forward(torch.from_numpy(bad).cuda(), hook_layers=[11], pooling="mean")
    for key in ("pooled", "features"):
        a, c = r[key].cpu().numpy(), clean[key].cpu().numpy()
        assert np.isnan(a[1]).all(), ("eat", key)
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[2], c[2]), ("eat", key)
    assert np.isnan(r["hooks"][11].cpu().numpy()[1]).all()
    enc.close()
    cfg = dict(synth.AVES_BASE_CFG, encoder_num_layers=2)
    aenc = AvesEncoder(cfg, synth.aves_state_dict(cfg))
    x = synth.noise_clips(3, 40000, seed=15)
    clean = aenc.forward(torch.from_numpy(x).cuda(), hook_layers=[1], want_features=True, want_pooled=True)
    bad = x.copy(); bad[1, 20000] = np.nan
    r = aenc.forward(torch.from_numpy(bad).cuda(), hook_layers=[1], want_features=True, want_pooled=True)
    for key in ("pooled", "features"):
        a, c = r[key].cpu().numpy(), clean[key].cpu().numpy()
        assert np.isnan(a[1]).all(), ("aves", key)
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[2], c[2]), ("aves", key)
    assert np.isnan(r["hooks"][1].cpu().numpy()[1]).all()
