"""Plugin/registry API behaviour on CPU (no forward): mirrors the reference's own unit tests
(tests/unittests/test_base_model.py, test_base_model_all_layers.py, test_api_load.py,
test_api_registry.py) for the parts of the contract the hot path sits behind."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

import avex_amd
from avex_amd import ModelBase, ModelSpec, registry, synth
from avex_amd._capi import AvexHipError


@pytest.fixture(scope="module")
def meta(golden_dir):
    with open(f"{golden_dir}/base_api.json") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def beats_cpu():
    spec = avex_amd.get_model_spec("esp_aves2_sl_beats_all").model_copy(deep=True)
    return avex_amd.build_model_from_spec(spec, "cpu", return_features_only=True)


def test_layer_map_is_bit_exact(beats_cpu, meta):
    assert {str(k): v for k, v in beats_cpu.get_model_layer_map().items()} == meta["layer_map"]
    assert beats_cpu.get_model_layers() == [meta["layer_map"][str(i)] for i in range(13)]


def test_hook_resolution_matches_reference(beats_cpu, meta):
    assert beats_cpu.register_hooks_for_layers([0, -1]) == meta["resolve_[0,-1]"]
    assert beats_cpu.register_hooks_for_layers(["all"]) == meta["resolve_all"]
    assert beats_cpu.register_hooks_for_layers(["last_layer"]) == meta["resolve_last_layer"]
    assert beats_cpu.register_hooks_for_layers([3, "backbone.encoder.layers.2.fc2", "all", -1]) == meta["resolve_mixed"]
    assert list(beats_cpu._hooks) == meta["resolve_mixed"] == beats_cpu._hook_layers
    beats_cpu.deregister_all_hooks()
    assert not beats_cpu._hooks and beats_cpu._hook_layers == meta["resolve_mixed"]
    beats_cpu.ensure_hooks_registered()
    assert list(beats_cpu._hooks) == meta["resolve_mixed"]
    beats_cpu.deregister_all_hooks()


def test_hook_errors_match_reference(beats_cpu, meta):
    errs = {"ValueError": ValueError, "TypeError": TypeError}
    with pytest.raises(errs[meta["errors"]["index_oob"]]):
        beats_cpu.register_hooks_for_layers([13])
    with pytest.raises(errs[meta["errors"]["neg_oob"]]):
        beats_cpu.register_hooks_for_layers([-14])
    with pytest.raises(errs[meta["errors"]["bool"]]):
        beats_cpu.register_hooks_for_layers([True])
    with pytest.raises(errs[meta["errors"]["unknown"]]):
        beats_cpu.register_hooks_for_layers(["backbone.nope"])
    beats_cpu.deregister_all_hooks()
    beats_cpu._hook_layers = []
    with pytest.raises(errs[meta["errors"]["no_hooks"]]):
        beats_cpu.extract_embeddings(torch.zeros(1, 16000), aggregation="mean")
    beats_cpu.register_hooks_for_layers([0])
    with pytest.raises(ValueError):
        beats_cpu.extract_embeddings(None)
    with pytest.raises(ValueError):
        beats_cpu.extract_embeddings(torch.zeros(1, 0))


def test_state_dict_keys_and_checkpoint_roundtrip(beats_cpu, tmp_path):
    ref = synth.beats_state_dict()
    keys = set(beats_cpu.state_dict())
    assert len(keys) == 254 and set(ref) <= keys
    assert keys - set(ref) == {"backbone.fbank.window", "backbone.fbank.mel_fb"}
    # shared relative-position table: one storage listed under every layer (backbone.py:100-103)
    sd = beats_cpu.state_dict()
    assert sd["backbone.encoder.layers.0.self_attn.relative_attention_bias.weight"].data_ptr() == \
        sd["backbone.encoder.layers.7.self_attn.relative_attention_bias.weight"].data_ptr()
    from safetensors.numpy import save_file
    small = {k: v for k, v in ref.items()}
    path = tmp_path / "ckpt.safetensors"
    save_file({k: np.ascontiguousarray(v) for k, v in small.items()}, str(path))
    m = avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=str(path), return_features_only=True)
    assert m.device == "cpu" and m.classifier is None
    for k in ("backbone.encoder.layers.5.fc2.weight", "backbone.post_extract_proj.bias",
              "backbone.encoder.pos_conv.0.parametrizations.weight.original1"):
        assert torch.equal(m.state_dict()[k], torch.from_numpy(ref[k]))
    # checkpoints without the backbone. prefix / wrapped .pt files load too (load.py:553-562)
    pt = tmp_path / "ckpt.pt"
    torch.save({"model": {k[len("backbone."):]: torch.from_numpy(v) for k, v in small.items()}}, pt)
    m2 = avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=str(pt), return_features_only=True)
    assert torch.equal(m2.state_dict()["backbone.encoder.layers.5.fc2.weight"], torch.from_numpy(ref["backbone.encoder.layers.5.fc2.weight"]))


def test_classifier_mode_recovers_num_classes(tmp_path):
    sd = {k: torch.from_numpy(v) for k, v in synth.beats_state_dict().items()}
    sd["classifier.weight"] = torch.zeros(37, 768)
    sd["classifier.bias"] = torch.zeros(37)
    pt = tmp_path / "clf.pt"
    torch.save({"model_state_dict": sd}, pt)
    m = avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=str(pt))
    assert m.num_classes == 37 and isinstance(m.classifier, nn.Linear) and m.classifier.out_features == 37
    m = avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path=str(pt), return_features_only=True)
    assert m.classifier is None and "classifier.weight" not in m.state_dict()


def test_load_model_errors():
    with pytest.raises(ValueError):
        avex_amd.load_model("definitely_not_a_model")
    with pytest.raises(TypeError):
        avex_amd.load_model(123)
    with pytest.raises(KeyError):
        avex_amd.build_model_from_spec(ModelSpec(name="not_a_class", pretrained=False), "cpu")
    with pytest.raises(FileNotFoundError):
        avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu", checkpoint_path="/nonexistent/x.safetensors")
    with pytest.raises(FileNotFoundError):      # default checkpoint is an hf:// URI: unreachable offline
        avex_amd.load_model("esp_aves2_sl_beats_all", device="cpu")
    with pytest.raises(KeyError):
        avex_amd.get_checkpoint_path("definitely_not_a_model")


def test_registry_lists_the_official_ids():
    ids = set(avex_amd.list_models())
    assert len(ids) >= 10 and {"esp_aves2_sl_beats_all", "esp_aves2_sl_beats_bio", "esp_aves2_naturelm_audio_v1_beats",
                               "esp_aves2_effnetb0_all", "esp_aves2_eat_all"} <= ids
    spec = avex_amd.get_model_spec("esp_aves2_sl_beats_all")
    assert spec.name == "beats" and spec.fine_tuned and spec.init_config["encoder_layers"] == 12
    assert spec.audio_config.representation == "raw" and spec.audio_config.target_length_seconds == 10
    assert avex_amd.get_model_spec("nope") is None
    assert avex_amd.get_checkpoint_path("esp_aves2_sl_beats_all").startswith("hf://EarthSpeciesProject/")
    d = avex_amd.describe_model("esp_aves2_naturelm_audio_v1_beats")
    assert d["model_spec"]["use_naturelm"] is True
    info = avex_amd.list_model_layers("esp_aves2_sl_beats_all")
    assert info["last_layer"] == "backbone.encoder.layers.11.fc2" and len(info["layers"]) == 13


def test_custom_model_class_plugin(tmp_path):
    """docs/custom_model_registration.md flow: register a class, a spec, build through the factory,
    use the generic hook machinery with an ordinary torch forward."""

    @avex_amd.register_model_class
    class TinyNet(ModelBase):
        name = "tiny_net_test"

        def __init__(self, device: str, audio_config=None, num_classes: int = 3, return_features_only: bool = False):
            super().__init__(device=device, audio_config=audio_config)
            self.body = nn.Sequential(nn.Linear(16, 8), nn.ReLU(), nn.Linear(8, 8))
            self.classifier = nn.Linear(8, num_classes)

        def forward(self, x, padding_mask=None):
            return self.classifier(self.body(x))

    assert "tiny_net_test" in avex_amd.list_model_classes()
    avex_amd.register_model("tiny_spec", ModelSpec(name="tiny_net_test", pretrained=False, device="cpu", model_id=None))
    m = avex_amd.build_model("tiny_spec", "cpu", num_classes=5, not_a_param=1)     # unknown kwargs are filtered
    assert m.classifier.out_features == 5
    assert m.get_model_layers() == ["body.0", "body.2", "classifier"]
    assert m.register_hooks_for_layers(["last_layer"]) == ["body.2"]
    x = torch.randn(4, 16)
    e = m.extract_embeddings(x, aggregation="mean")
    assert e.shape == (4, 8) and torch.allclose(e, m.body(x))
    m.register_hooks_for_layers(["all"])
    e = m.extract_embeddings({"raw_wav": x}, aggregation="none")
    assert isinstance(e, list) and [tuple(t.shape) for t in e] == [(4, 8), (4, 8), (4, 5)]
    assert m.extract_embeddings(x, aggregation="mean").shape == (4, 21)
    assert not m._hook_outputs                       # cleared after extraction
    yml = tmp_path / "my_tiny.yml"
    yml.write_text("model_spec:\n  name: tiny_net_test\n  pretrained: false\n  device: cpu\n")
    m2 = avex_amd.load_model(str(yml), device="cpu", return_features_only=True)
    assert isinstance(m2, TinyNet) and avex_amd.get_model_spec("my_tiny") is not None


def test_beats_forward_needs_gpu(beats_cpu):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(AvexHipError):
        beats_cpu(torch.zeros(1, 16000))


def test_padding_mask_geometry(beats_cpu):
    """forward_padding_mask twice: 32000 samples -> 198 frames (drop 32000 % 198) -> 96 tokens (drop 198 % 96)."""
    from oracle import beats_oracle as O
    pm = torch.zeros(2, 32000, dtype=torch.bool); pm[1, 16000:] = True
    a = beats_cpu.forward_padding_mask(96, beats_cpu.forward_padding_mask(198, pm))
    b = O.forward_padding_mask(96, O.forward_padding_mask(198, pm.numpy()))
    assert a.shape == (2, 96) and np.array_equal(a.numpy(), b) and a[1].sum() > 40 and not a[0].any()


def test_unsupported_variants_fail_loudly():
    # the reference's own refusals, with its exception types: an activation get_activation_fn does not know (modules.py:237) and
    # deep_norm together with layer_norm_first (beats.py:275).  Pre-LN, the other activations, glu and conv_bias are built (ABI 6).
    with pytest.raises(RuntimeError):
        avex_amd.beats_model.Model(device="cpu", init_config=dict(synth.BEATS_BASE_CFG, activation_fn="swish"))
    with pytest.raises(AssertionError):
        avex_amd.beats_model.Model(device="cpu", init_config=dict(synth.BEATS_BASE_CFG, layer_norm_first=True, deep_norm=True))
    avex_amd.beats_model.Model(device="cpu", init_config=dict(synth.BEATS_BASE_CFG, layer_norm_first=True, deep_norm=False))
    with pytest.raises(FileNotFoundError):
        avex_amd.beats_model.Model(device="cpu", pretrained=True)


def test_extraction_loop_semantics_cpu():
    """Harness counterpart (reference: evaluation/embedding_utils.py:26-144) with a stand-in model on the CPU: batch schema,
    single tensor / list / dict returns, stacking order, embedding_dims, ValueError on an empty loader, hooks deregistered and
    disable_layerdrop restored in `finally` even when a batch raises."""
    import pytest
    import torch
    from avex_amd.extraction import extract_embeddings_in_memory

    class Fake:
        def __init__(self, mode):
            self.mode, self.disable_layerdrop, self.registered, self.deregistered, self.calls = mode, False, None, 0, []

        def register_hooks_for_layers(self, layers):
            self.registered = list(layers)
            return ["L0", "L1"][: len(layers)]

        def deregister_all_hooks(self):
            self.deregistered += 1

        def extract_embeddings(self, x, aggregation="none"):
            masked = isinstance(x, dict)
            wav = x["raw_wav"] if masked else x
            self.calls.append((masked, aggregation, self.disable_layerdrop))
            base = wav.sum(dim=1, keepdim=True)
            if self.mode == "tensor":
                return base.repeat(1, 4)
            if self.mode == "list":
                return [base.repeat(1, 4), base.repeat(1, 3) * 2]
            if self.mode == "dict":
                return {"a": base.repeat(1, 2), "b": base.repeat(1, 5)}
            raise RuntimeError("boom")

    batches = [{"raw_wav": torch.full((2, 8), float(i + 1)), "padding_mask": torch.zeros(2, 8, dtype=torch.bool) if i % 2 else None,
                "label": torch.tensor([i, i + 10])} for i in range(3)]
    for b in batches:
        if b["padding_mask"] is None:
            del b["padding_mask"]
    m = Fake("tensor")
    emb, labels, dims = extract_embeddings_in_memory(m, batches, [0], "cpu", aggregation="mean", disable_layerdrop=True)
    assert list(emb) == ["L0"] and emb["L0"].shape == (6, 4) and dims == [(4,)]
    assert labels.tolist() == [0, 10, 1, 11, 2, 12]
    assert emb["L0"][:, 0].tolist() == [8.0, 8.0, 16.0, 16.0, 24.0, 24.0]                 # batch order kept
    assert [c[0] for c in m.calls] == [False, True, False] and all(c[1] == "mean" and c[2] is True for c in m.calls)
    assert m.deregistered == 1 and m.disable_layerdrop is False and m.registered == [0]
    m = Fake("list")
    emb, _, dims = extract_embeddings_in_memory(m, batches, [0, -1], "cpu")
    assert list(emb) == ["L0", "L1"] and dims == [(4,), (3,)]
    m = Fake("dict")
    emb, _, dims = extract_embeddings_in_memory(m, batches, ["all"], "cpu")
    assert list(emb) == ["a", "b"] and dims == [(2,), (5,)]
    m = Fake("tensor")
    with pytest.raises(ValueError):
        extract_embeddings_in_memory(m, [], [0], "cpu")
    assert m.deregistered == 1
    m = Fake("raise")
    with pytest.raises(RuntimeError):
        extract_embeddings_in_memory(m, batches, [0], "cpu", disable_layerdrop=True)
    assert m.deregistered == 1 and m.disable_layerdrop is False


def test_streaming_extraction_cache_layout_cpu(tmp_path):
    """The HDF5 cache writer (reference: evaluation/embedding_utils.py:147-161, 164-346, 349-822) against an in-memory stand-in for
    h5py (not installed here): dataset names / shapes / dtypes / chunking, row order, and every attribute the reference's reader
    (`_load_metadata`, embedding_utils.py:1085-1101) looks for; hooks deregistered afterwards."""
    import numpy as np
    import torch
    from avex_amd.extraction import extract_embeddings_streaming, write_embedding_metadata

    class Dset:
        def __init__(self, shape, dtype, **kw):
            self.a, self.kw = np.zeros(shape, dtype), kw
            self.shape = tuple(shape)

        def __setitem__(self, k, v):
            self.a[k] = v

        def __getitem__(self, k):
            return self.a[k]

    class File(dict):
        opened = []

        def __init__(self, path, mode):
            super().__init__()
            self.attrs, self.path, self.mode = {}, path, mode
            File.opened.append(self)

        def create_dataset(self, name, shape, maxshape, dtype, chunks, **kw):
            self[name] = Dset(shape, dtype, maxshape=maxshape, chunks=chunks, **kw)
            return self[name]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class H5:
        pass
    H5.File = File

    class Fake:
        deregistered = 0

        def register_hooks_for_layers(self, layers):
            return ["backbone.post_extract_proj", "backbone.encoder.layers.11.fc2"][: len(layers)]

        def deregister_all_hooks(self):
            self.deregistered += 1

        def extract_embeddings(self, x, aggregation="none"):
            wav = x["raw_wav"] if isinstance(x, dict) else x
            base = wav.sum(dim=1, keepdim=True)
            if aggregation == "none":
                return [base[:, None, :].repeat(1, 5, 4), base[:, None, :].repeat(1, 5, 3) * 2]
            return base.repeat(1, 8)

    class Loader(list):
        pass
    batches = Loader({"raw_wav": torch.full((2 if i < 2 else 1, 8), float(i + 1)), "label": torch.tensor([i, i + 10][: 2 if i < 2 else 1])} for i in range(3))
    batches.dataset = list(range(5))
    m = Fake()
    out = tmp_path / "cache" / "emb.h5"
    dims = extract_embeddings_streaming(m, batches, [0], "cpu", out, chunk_size=4, compression="gzip", compression_level=4, aggregation="mean", h5_module=H5)
    f = File.opened[-1]
    assert dims == [(8,)] and f.mode == "w" and out.parent.is_dir()
    assert set(f) == {"embeddings_backbone.post_extract_proj", "labels"}
    d = f["embeddings_backbone.post_extract_proj"]
    assert d.shape == (5, 8) and d.a.dtype == np.float32 and d.kw["chunks"] == (4, 8) and d.kw["maxshape"] == (None, 8)
    assert d.kw["compression"] == "gzip" and d.kw["compression_opts"] == 4
    assert d.a[:, 0].tolist() == [8.0, 8.0, 16.0, 16.0, 24.0]
    assert f["labels"].a.dtype == np.int64 and f["labels"].a.tolist() == [0, 10, 1, 11, 2]
    assert f.attrs == {"num_labels": 5, "embedding_aggregation": "mean", "aggregation": "mean", "stored_embedding_rank": [1],
                       "layer_names": ["backbone.post_extract_proj"], "embedding_dims": ["(8,)"], "multi_layer": True,
                       "extraction_complete": True, "skipped_batches": 0}
    assert m.deregistered == 1
    # two layers, unpooled, no compression (lzf takes no level): per-layer datasets of their own rank
    dims = extract_embeddings_streaming(m, batches, [0, -1], "cpu", out, chunk_size=100, compression="lzf", aggregation="none", h5_module=H5)
    f = File.opened[-1]
    assert dims == [(5, 4), (5, 3)]
    assert f["embeddings_backbone.encoder.layers.11.fc2"].shape == (5, 5, 3) and f["embeddings_backbone.encoder.layers.11.fc2"].kw["chunks"] == (5, 5, 3)
    assert "compression_opts" not in f["embeddings_backbone.post_extract_proj"].kw
    assert f.attrs["stored_embedding_rank"] == [2, 2] and f.attrs["embedding_dims"] == ["(5, 4)", "(5, 3)"]
    store = File("x", "w")
    write_embedding_metadata(store, aggregation="max", layer_names=["a"], embedding_dims=[(3, 4)], multi_layer=False)
    assert store.attrs["stored_embedding_rank"] == [2] and store.attrs["multi_layer"] is False and store.attrs["aggregation"] == "max"


def test_aves_class_contract_cpu():
    """AVES mirror (reference: avex/models/aves_model.py:62-262) without a GPU: torchaudio wav2vec2 key names (210 tensors),
    hookable layers = the 12 output_dense modules, name/index resolution, prefix-less and old-style weight_norm checkpoints
    load, forward refuses to run on the CPU."""
    import numpy as np
    import pytest
    import torch
    import avex_amd
    from avex_amd import synth
    from avex_amd._capi import AvexHipError
    from avex_amd.aves_model import Model
    assert "aves" in avex_amd.list_model_classes()
    m = Model(device="cpu")
    sd = synth.aves_state_dict()
    assert set(m.state_dict()) == set(sd) and len(sd) == 210
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    bare = {k[len("model."):]: torch.from_numpy(v) for k, v in sd.items()}            # as torchaudio saves it
    bare["encoder.transformer.pos_conv_embed.conv.weight_g"] = bare.pop("encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original0")
    bare["encoder.transformer.pos_conv_embed.conv.weight_v"] = bare.pop("encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original1")
    m.load_state_dict(bare)
    layers = m.register_hooks_for_layers(["all"])
    assert layers == [f"model.encoder.transformer.layers.{i}.feed_forward.output_dense" for i in range(12)]
    assert m.register_hooks_for_layers([-1, 0]) == [layers[-1], layers[0]]
    with pytest.raises(ValueError):
        m.register_hooks_for_layers(["model.encoder.nope"])
    m.deregister_all_hooks()
    with pytest.raises(ValueError):
        m.extract_embeddings(torch.zeros(1, 16000))                                    # no hooks registered
    with pytest.raises((AvexHipError, RuntimeError)):
        m(torch.zeros(1, 16000))
    with pytest.raises(FileNotFoundError):
        Model(device="cpu", pretrained=True)


def test_efficientnet_class_contract_cpu():
    """EfficientNet mirror (reference: avex/models/efficientnet.py:21-322) without a GPU: torchvision efficientnet_b0 key names,
    the 17 hookable convolutions (stem, every block.3.0, head), prefix-less checkpoints, variants, refusal to run on the CPU."""
    import pytest
    import torch
    import avex_amd
    from avex_amd import synth
    from avex_amd._capi import AvexHipError
    from avex_amd.efficientnet import Model
    assert "efficientnet" in avex_amd.list_model_classes()
    m = Model(device="cpu", return_features_only=True)
    sd = synth.effnet_b0_state_dict()
    feat_keys = {k for k in m.state_dict() if k.startswith("model.features.")}
    assert feat_keys == set(sd) and len(sd) == 358
    assert {"model.classifier.1.weight", "model.classifier.1.bias"} <= set(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.load_state_dict({k[len("model."):]: torch.from_numpy(v) for k, v in sd.items()}, strict=False)     # as torchvision saves it
    layers = m.register_hooks_for_layers(["all"])
    assert len(layers) == 17 and layers[0] == "model.features.0.0" and layers[-1] == "model.features.8.0"
    assert layers[1] == "model.features.2.0.block.3.0" and all(n.endswith(".block.3.0") for n in layers[1:-1])
    assert m.register_hooks_for_layers(["last_layer"]) == ["model.features.8.0"] or len(m.register_hooks_for_layers(["last_layer"])) == 1
    m.deregister_all_hooks()
    assert Model(device="cpu", num_classes=7).model.classifier[1].out_features == 7
    b1 = Model(device="cpu", efficientnet_variant="b1", return_features_only=True)     # efficientnet.py:64-66: same widths, deeper stages
    assert [len(st) for st in list(b1.model.features)[1:8]] == [2, 3, 3, 4, 4, 5, 2]
    assert {k for k in b1.state_dict() if k.startswith("model.features.")} == set(synth.effnet_b0_state_dict(stages=synth.EFFNET_B1_STAGES))
    assert len(b1.register_hooks_for_layers(["all"])) == 23                             # stem + 21 `block.3.0` projections (stage 1 has no expansion, its projection is block.2.0) + head
    with pytest.raises(ValueError):
        Model(device="cpu", efficientnet_variant="b9")
    with pytest.raises(FileNotFoundError):
        Model(device="cpu", pretrained=True)
    with pytest.raises((AvexHipError, RuntimeError)):
        m(torch.zeros(1, 3, 16, 16))
    spec = avex_amd.get_model_spec("esp_aves2_effnetb0_all")
    assert spec.name == "efficientnet" and spec.audio_config.n_fft == 800 and spec.audio_config.representation == "mel_spectrogram"


def test_bench_effnet_algorithmic_bytes():
    import os
    """bench.py's C5 roofline prices the EfficientNet leg on ALGORITHMIC bytes: every layer's input read once and its output written once at
    2 bytes and the real channel counts, the frontend's fp32 wav in / image out (DESIGN.md section 4, "Other BASELINE configs"): pin the
    figure for B0 on a 128 x 1001 mel image, and its pieces for a one-block toy stack."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from avex_amd.synth import EFFNET_B0_STAGES
    b = bench.effnet_algorithmic_bytes(EFFNET_B0_STAGES, 128, 1001, 160000)
    assert b == 71942656
    # toy: 8 x 8 image, stem 4 channels (-> 4 x 4), one expand-free 3x3 stride-1 block 4 -> 4 channels with a residual, head 8 channels
    toy = bench.effnet_algorithmic_bytes([(1, 3, 1, 4, 4, 1)], 8, 8, 100, stem=4, head=8)
    want = 100 * 4 + 64 * 4 + 64 * 4 + 16 * 4 * 2 + (16 + 16) * 4 * 2 + 16 * (4 + 4) * 2 + 16 * 4 * 2 + 16 * (4 + 2 * 8) * 2 + 8 * 4
    assert toy == want
