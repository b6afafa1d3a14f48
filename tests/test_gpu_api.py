"""Drop-in API on the GPU (`-m gpu`): load_model -> forward / hooks / extract_embeddings, checked
against the goldens the real reference produced through the same public calls."""
import json

import numpy as np
import pytest
import torch

import avex_amd
from _util import rel_l2
from avex_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-3      # BASELINE.json: pooled 768-d embedding within 1e-3 relative of the reference CPU path


@pytest.fixture(scope="module")
def model(built_lib, tmp_path_factory):
    from safetensors.numpy import save_file
    path = tmp_path_factory.mktemp("ckpt") / "synthetic_beats.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0).items()}, str(path))
    m = avex_amd.load_model("esp_aves2_sl_beats_all", device="cuda", checkpoint_path=str(path), return_features_only=True)
    return m.eval()


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(f"{golden_dir}/base_api.npz")


def test_forward_features(model, g):
    x = torch.from_numpy(synth.noise_clips(1, 160000, seed=0))          # CPU tensor in: process_audio moves it
    f = model(x)
    assert f.shape == (1, 496, 768) and f.is_cuda and f.dtype == torch.float32
    assert rel_l2(f.mean(1).cpu().numpy(), g["b1.pooled"]) < TOL
    assert model.device == "cuda"


def test_extract_embeddings_aggregations(model, g, golden_dir):
    with open(f"{golden_dir}/base_api.json") as fh:
        meta = json.load(fh)
    assert model.register_hooks_for_layers([0, -1]) == meta["resolve_[0,-1]"]
    x = torch.from_numpy(synth.noise_clips(2, 32000, seed=5)).cuda()
    for agg in ("mean", "max", "cls_token"):
        e = model.extract_embeddings(x, aggregation=agg)
        assert e.shape == (2, 1536)
        assert rel_l2(e.cpu().numpy(), g[f"agg.{agg}"]) < (TOL if agg == "mean" else 4 * TOL)   # max/cls are frame-level
    lst = model.extract_embeddings(x, aggregation="none")
    assert [list(t.shape) for t in lst] == meta["agg_none_shapes"]
    assert rel_l2(lst[0][:, ::8].cpu().numpy(), g["agg.none0_tok8"]) < 4 * TOL
    assert rel_l2(lst[1][:, ::8].cpu().numpy(), g["agg.none1_tok8"]) < 4 * TOL
    # one layer -> a tensor, not a list
    model.register_hooks_for_layers(["last_layer"])
    one = model.extract_embeddings(x, aggregation="none")
    assert isinstance(one, torch.Tensor) and one.shape == (2, 96, 768)
    assert not model._hook_outputs


def test_all_layers_mean(model, g):
    names = model.register_hooks_for_layers(["all"])
    assert len(names) == 13
    x = torch.from_numpy(synth.noise_clips(4, 160000, seed=0)).cuda()
    e = model.extract_embeddings(x, aggregation="mean")
    assert e.shape == (4, 13 * 768)
    assert rel_l2(e.cpu().numpy(), g["b4.all_mean"]) < TOL


def test_dict_input_with_padding_mask(model, g):
    model.register_hooks_for_layers([0, -1])
    x = torch.from_numpy(synth.noise_clips(2, 32000, seed=5)).cuda()
    pm = torch.zeros(2, 32000, dtype=torch.bool); pm[1, 16000:] = True
    e = model.extract_embeddings({"raw_wav": x, "padding_mask": pm.cuda()}, aggregation="mean")
    assert rel_l2(e.cpu().numpy(), g["mask.mean"]) < TOL
    f = model(x, pm)
    assert rel_l2(f[:, ::8].cpu().numpy(), g["mask.features_tok8"]) < 4 * TOL


def test_user_forward_hook_fires(model):
    """A hook registered directly on a tap module (not via register_hooks_for_layers) is delivered too."""
    model.deregister_all_hooks()
    seen = {}
    mod = model.get_submodule("backbone.encoder.layers.4.fc2")
    h = mod.register_forward_hook(lambda m, i, o: seen.setdefault("out", o))
    x = torch.from_numpy(synth.noise_clips(1, 16000, seed=2)).cuda()
    model(x)
    h.remove()
    assert seen["out"].shape == (1, 48, 768)
    model(x)                                    # no hooks -> no taps requested, still fine


def test_tones_and_regression_inputs(model, g):
    x = torch.from_numpy(synth.tone_clips(16000)).cuda()
    f = model(x)
    assert rel_l2(f.mean(1).cpu().numpy(), g["tone.pooled"]) < TOL


def test_classifier_mode_and_reload(model, tmp_path):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sd["classifier.weight"] = torch.from_numpy(synth.normal("clfw", (5, 768), 0.05))
    sd["classifier.bias"] = torch.from_numpy(synth.normal("clfb", (5,), 0.05))
    pt = tmp_path / "clf.pt"
    torch.save({"model_state_dict": sd}, pt)
    m = avex_amd.load_model("esp_aves2_sl_beats_all", device="cuda", checkpoint_path=str(pt)).eval()
    x = torch.from_numpy(synth.noise_clips(2, 32000, seed=5)).cuda()
    logits = m(x)
    assert logits.shape == (2, 5)
    feats = model(x)
    ref = feats.mean(1) @ sd["classifier.weight"].cuda().T + sd["classifier.bias"].cuda()
    assert torch.allclose(logits, ref, atol=1e-4)
    # masked mean pooling when a padding mask is given (beats_model.py:269-273)
    pm = torch.zeros(2, 32000, dtype=torch.bool); pm[1, 16000:] = True
    lm = m(x, pm)
    fm = model(x, pm)
    fp = model.forward_padding_mask(96, model.forward_padding_mask(198, pm)).cuda()
    keep = (~fp).unsqueeze(-1).float()
    ref = ((fm * keep).sum(1) / keep.sum(1)) @ sd["classifier.weight"].cuda().T + sd["classifier.bias"].cuda()
    assert torch.allclose(lm, ref, atol=1e-4)
    # in-place weight edit + refresh is picked up by the HIP handle
    with torch.no_grad():
        m.backbone.encoder.layers[11].final_layer_norm.bias.add_(1.0)
    m.refresh_weights()
    assert not torch.allclose(m(x), logits)


def test_frontend_namespace(g, golden_dir):
    from avex_amd import preprocessing
    fb = np.load(f"{golden_dir}/fbank.npz")
    x = torch.from_numpy(synth.noise_clips(2, 16000, seed=0)).cuda()
    y = preprocessing.BatchedFbank()(x * 2 ** 15)
    assert y.shape == (2, 98, 128) and float(np.abs(y.cpu().numpy() - fb["noise16k"]).max()) < 1e-3
    z = preprocessing.beats_preprocess(x)
    assert torch.allclose(z, (y - 15.41663) / (2 * 6.55582), atol=2e-5)


def test_extraction_loop_on_gpu(model, g):
    """The harness loop (reference: embedding_utils.py:26-144) over a Collater-style loader with the real model: pinned/async
    prefetch and the plain synchronous path give the same rows, in loader order, and the golden pooled embedding comes out."""
    from avex_amd.extraction import extract_embeddings_in_memory
    x = synth.noise_clips(4, 160000, seed=0)
    pad = torch.zeros(2, 160000, dtype=torch.bool); pad[1, 80000:] = True
    loader = [{"raw_wav": torch.from_numpy(x[:1]), "label": torch.tensor([7])},
              {"raw_wav": torch.from_numpy(x[1:3]), "padding_mask": pad, "label": torch.tensor([1, 2])},
              {"raw_wav": torch.from_numpy(x[3:4]), "label": torch.tensor([3])}]
    a, la, dims = extract_embeddings_in_memory(model, loader, [0, -1], "cuda", aggregation="mean", prefetch=True)
    b, lb, _ = extract_embeddings_in_memory(model, loader, [0, -1], "cuda", aggregation="mean", prefetch=False)
    assert la.tolist() == lb.tolist() == [7, 1, 2, 3]
    assert len(a) == 1 and dims == [(1536,)]                    # mean aggregation concatenates the two taps: one entry, like the reference
    (name, ea), (_, eb) = next(iter(a.items())), next(iter(b.items()))
    assert name == "backbone.post_extract_proj" and ea.shape == (4, 1536) and not ea.is_cuda
    assert torch.equal(ea, eb)
    model.register_hooks_for_layers([0, -1])
    try:
        row0 = model.extract_embeddings(torch.from_numpy(x[:1]), aggregation="mean").cpu()
    finally:
        model.deregister_all_hooks()
    assert torch.equal(ea[:1], row0)
    assert not model._hooks                                     # loop deregistered its hooks


def test_aves_model_on_gpu(built_lib):
    """AVES mirror end to end on the GPU vs the NumPy restatement (2 transformer layers, real widths): features, hook taps
    through extract_embeddings with every aggregation the reference offers."""
    from oracle import aves_oracle as AO
    from avex_amd.aves_model import Model
    cfg = dict(synth.AVES_BASE_CFG, encoder_num_layers=2)
    sd = synth.aves_state_dict(cfg)
    m = Model(device="cuda", init_config={"encoder_num_layers": 2}).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = synth.noise_clips(2, 32000, seed=47)
    ref, taps = AO.aves_forward(x, sd, cfg)
    f = m(torch.from_numpy(x))
    assert f.shape == ref.shape == (2, 99, 768) and f.is_cuda
    assert rel_l2(f.mean(1).cpu().numpy(), ref.mean(1)) < 2e-3
    names = m.register_hooks_for_layers(["all"])
    t0, t1 = taps[names[0]], taps[names[1]]
    e = m.extract_embeddings({"raw_wav": torch.from_numpy(x)}, aggregation="mean")
    assert e.shape == (2, 1536) and rel_l2(e.cpu().numpy(), np.concatenate([t0.mean(1), t1.mean(1)], 1)) < 3e-3
    e = m.extract_embeddings(torch.from_numpy(x), aggregation="max")
    assert rel_l2(e.cpu().numpy(), np.concatenate([t0.max(1), t1.max(1)], 1)) < 1e-2
    e = m.extract_embeddings(torch.from_numpy(x), aggregation="none")
    assert isinstance(e, list) and len(e) == 2 and e[1].shape == (2, 99, 768)
    m.deregister_all_hooks()


def test_efficientnet_model_on_gpu(built_lib, tmp_path):
    """EfficientNet mirror end to end through load_model with a local checkpoint: waveform -> GPU mel frontend -> B0 features,
    vs the NumPy restatements of both stages (1 s clips keep the CPU side short); hook taps through extract_embeddings."""
    from safetensors.numpy import save_file
    from oracle import beats_oracle as BO
    from oracle import effnet_oracle as EO
    sd = synth.effnet_b0_state_dict()
    path = tmp_path / "effnet.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items() if not k.endswith("num_batches_tracked")}, str(path))
    m = avex_amd.load_model("esp_aves2_effnetb0_all", device="cuda", checkpoint_path=str(path), return_features_only=True).eval()
    x = synth.noise_clips(2, 16000, seed=71)
    mel = BO.audio_processor(x, n_fft=800, hop=160)
    ref, taps = EO.effnet_features(mel, sd, synth.EFFNET_B0_STAGES)
    f = m(torch.from_numpy(x))
    assert f.shape == ref.shape == (2, 1280, 4, 4) and f.is_cuda
    assert rel_l2(f.cpu().numpy(), ref) < 2e-2
    names = m.register_hooks_for_layers([0, -1])
    assert names == ["model.features.0.0", "model.features.8.0"]
    e = m.extract_embeddings({"raw_wav": torch.from_numpy(x)}, aggregation="mean")
    want = np.concatenate([taps[names[0]].mean(-1).reshape(2, -1), taps[names[1]].mean(-1).reshape(2, -1)], 1)
    assert e.shape == want.shape == (2, 32 * 64 + 1280 * 4) and rel_l2(e.cpu().numpy(), want) < 2e-2
    lst = m.extract_embeddings(torch.from_numpy(x), aggregation="none")
    assert isinstance(lst, list) and lst[0].shape == (2, 32, 64, 51)
    m.deregister_all_hooks()


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


@pytest.mark.parametrize("variant", ["preln_relu_convbias", "postln_glu_nogate"])
def test_config_space_variants_through_the_model_class(built_lib, golden_dir, variant):
    """A BEATsConfig the official checkpoints do not use, through the plugin class: init_config -> parameter tree with the reference's
    key names (fc1.linear.* for glu, patch_embedding.bias) -> load_state_dict -> forward / hooks, against the real reference's outputs
    (tests/golden/make_variant_goldens.py)."""
    gv = np.load(f"{golden_dir}/variants.npz")
    cfg = synth.BEATS_VARIANTS[variant]
    m = avex_amd.beats_model.Model(device="cuda", init_config=cfg, return_features_only=True).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth.beats_state_dict(cfg, seed=3).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"backbone.fbank.window", "backbone.fbank.mel_fb"}
    x = torch.from_numpy(synth.noise_clips(2, 32000, seed=13)).cuda()
    f = m(x)
    assert rel_l2(f.mean(1).cpu().numpy(), gv[f"{variant}.pooled"]) < TOL
    assert rel_l2(f[:, ::3].cpu().numpy(), gv[f"{variant}.features_tok3"]) < 4 * TOL
    assert m.register_hooks_for_layers(["last_layer"]) == ["backbone.encoder.layers.1.fc2"]
    e = m.extract_embeddings(x, aggregation="mean")
    assert rel_l2(e.cpu().numpy(), gv[f"{variant}.fc2.1_mean"]) < 2 * TOL
    m.deregister_all_hooks()
