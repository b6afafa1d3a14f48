"""The reference-side binding of INTEGRATION.md section 2, as recorded against the REAL reference.

``tests/golden/binding_contract.json`` is the outcome of ``tests/golden/make_binding_contract.py``: the documented module executed
verbatim, then the reference's own ``register_model_class`` / ``build_model_from_spec`` / ``load_model`` / ``list_model_layers`` /
``build_probe_from_config`` driven over the four HIP classes (factory.py:108-166, load.py:152-311,521-570, registry.py:600-710,
probes/base_probes.py:23-195).  These tests assert that record, tie it to the current text of INTEGRATION.md and to what the
mirror's own registry says today, and -- in the development container, where /root/reference exists -- run the script again.
"""
import hashlib
import inspect
import json
import os
import re
import subprocess
import sys
from typing import Literal, Optional

import pytest
from pydantic import BaseModel

import avex_amd
from avex_amd import registry
from avex_amd.base_model import coerce_audio_config
from avex_amd.configs import AudioConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIRROR = {"beats_hip": "beats", "eat_hf_hip": "eat_hf", "efficientnet_hip": "efficientnet", "aves_hip": "aves"}


@pytest.fixture(scope="module")
def contract(golden_dir):
    with open(os.path.join(golden_dir, "binding_contract.json")) as f:
        return json.load(f)


def test_fixture_is_of_the_documented_block(contract):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# avex/models/hip_backends\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md lost its reference-side module block"
    assert hashlib.sha256(m.group(1).encode()).hexdigest() == contract["integration_block_sha256"], \
        "INTEGRATION.md section 2 changed after the contract was recorded: re-run tests/golden/make_binding_contract.py"


def test_every_class_registers_builds_and_loads_through_the_reference(contract):
    assert contract["registered_keys"] == sorted(MIRROR)
    for key, fam in contract["families"].items():
        assert fam["mro"][1].startswith("avex_amd."), fam["mro"]                 # this package's methods win the MRO
        assert fam["mro"][3] == "avex.models.base_model.ModelBase"
        b = fam["build"]
        assert b["isinstance_reference_ModelBase"] is True                        # registry.py:695
        assert b["device_attr"] == "cpu"
        assert b["init_keywords"]["audio_config"]["__type__"] == "avex.configs.AudioConfig"       # factory.py:132-143
        ld = fam["load"]
        assert ld["tensors_equal_after_load"] == ld["checkpoint_tensors"] > 100, (key, ld)       # _load_checkpoint matched every tensor
        assert ld["model_state_dict_keys"] >= ld["checkpoint_tensors"]
    off = contract["official_id_through_replaced_class"]
    assert off["class_mro_has_hip"] and off["tensors_equal_after_load"] == 252
    assert off["layers"] == contract["families"]["beats_hip"]["layers"]["layers"]


def test_factory_keywords_are_constructor_parameters(contract):
    """What the reference's factory passed after its signature filter is what the mirror's classes accept (and nothing more)."""
    for key, fam in contract["families"].items():
        cls = registry.get_model_class(MIRROR[key])
        params = set(inspect.signature(cls.__init__).parameters)
        for rec in (fam["build"]["init_keywords"], fam["load"]["init_keywords"]):
            assert set(rec) <= params, (key, set(rec) - params)
        assert fam["supports_return_features_only"] == ("return_features_only" in params)
    assert contract["families"]["aves_hip"]["supports_return_features_only"] is False          # aves_model.py:74-81
    eat = contract["families"]["eat_hf_hip"]["build"]["init_keywords"]
    assert not {"eat_norm_mean", "eat_norm_std", "model_id"} & set(eat)                         # dropped by the filter, as for the built-in


def test_layer_maps_equal_the_mirror(contract):
    specs = {"beats_hip": "esp_aves2_sl_beats_all", "eat_hf_hip": "esp_aves2_eat_all", "efficientnet_hip": "esp_aves2_effnetb0_all"}
    for key, fam in contract["families"].items():
        if key == "aves_hip":
            m = registry.get_model_class("aves")(device="cpu", num_classes=None,
                                                 audio_config=AudioConfig(representation="raw", normalize=False, target_length_seconds=10))
        else:
            m = avex_amd.build_model_from_spec(avex_amd.get_model_spec(specs[key]).model_copy(deep=True), "cpu", return_features_only=True)
        info = avex_amd.list_model_layers(m)
        assert info["layers"] == fam["layers"]["layers"], key
        assert info["last_layer"] == fam["layers"]["last_layer"]
        assert m.register_hooks_for_layers([0, -1]) == fam["hooks_0_m1"]
        assert len(m.register_hooks_for_layers(["all"])) == fam["hooks_all_count"]
        assert m.register_hooks_for_layers(["last_layer"]) == fam["hooks_last_layer"]
        m.deregister_all_hooks()
        ap = m.audio_processor
        tl = int(ap.target_length_seconds * ap.sr) if getattr(ap, "target_length_seconds", None) else int(ap.target_length)
        assert tl == fam["build"]["target_length_samples"]


def test_probe_factory_call_sequence(contract):
    for key, fam in contract["families"].items():
        for name in ("probe_linear_mean", "probe_linear_two_layers", "probe_attention_none"):
            if name not in fam:
                continue
            p = fam[name]
            assert len(p["construction_calls"]) == 1 and len(p["forward_calls"]) == 1
            c, f = p["construction_calls"][0], p["forward_calls"][0]
            assert c["input_shape"] == [1, fam["build"]["target_length_samples"]] and c["input_dtype"] == "float32"     # base_probes.py:121-125
            assert c["keywords"] == {"aggregation": p["probe_config"]["aggregation"]}
            assert set(f["keywords"]) == {"padding_mask", "aggregation", "freeze_backbone"}                           # base_probes.py:184-189
            assert c["hooks_registered"] == p["resolved_layers"]
            assert p["model_training_after"] is False and p["requires_grad_any"] is False                               # frozen backbone (base_probes.py:47-53)
            assert p["logits_shape"] == [2, 5]
            if name != "probe_attention_none":
                assert p["inferred_dim"] == (5120 if key == "efficientnet_hip" else 768) * len(p["resolved_layers"])
            # extract_embeddings of the mirror takes exactly these keywords
            sig = inspect.signature(registry.get_model_class(MIRROR[key]).extract_embeddings).parameters
            assert set(f["keywords"]) <= set(sig), key


# ------------------------------------------------------------------ the defect of round 2: a foreign AudioConfig
class _ForeignAudioConfig(BaseModel):
    """Stands in for avex.configs.AudioConfig on a machine without the reference: same fields, another class."""
    sample_rate: int = 16000
    n_fft: int = 2048
    hop_length: Optional[int] = None
    win_length: Optional[int] = None
    window: Literal["hann", "hamming"] = "hann"
    n_mels: int = 128
    representation: Literal["spectrogram", "mel_spectrogram", "raw"] = "mel_spectrogram"
    normalize: bool = True
    target_length_seconds: Optional[int] = None
    window_selection: Literal["random", "center"] = "random"
    center: bool = True
    extra_config: Optional[dict] = None
    a_field_of_a_later_reference_version: int = 7


class _PlainAudioConfig:
    sample_rate = 8000
    representation = "raw"
    target_length_seconds = 3


def test_coerce_audio_config_forms():
    assert coerce_audio_config(None) is None
    own = AudioConfig(representation="raw")
    assert coerce_audio_config(own) is own
    got = coerce_audio_config(_ForeignAudioConfig(representation="raw", normalize=False, target_length_seconds=10))
    assert isinstance(got, AudioConfig) and got.representation == "raw" and got.target_length_seconds == 10 and got.normalize is False
    got = coerce_audio_config(_PlainAudioConfig())
    assert got.sample_rate == 8000 and got.representation == "raw" and got.n_fft == 2048
    assert coerce_audio_config({"sample_rate": 32000}).sample_rate == 32000
    with pytest.raises(Exception):
        coerce_audio_config({"sample_rate": 32000, "unknown": 1})               # a mapping is validated strictly
    with pytest.raises(TypeError):
        coerce_audio_config(object())
    with pytest.raises(Exception):
        coerce_audio_config(_ForeignAudioConfig(sample_rate=16000).model_copy(update={"sample_rate": -1}))


@pytest.mark.parametrize("key", ["beats", "eat_hf", "efficientnet", "aves"])
def test_every_class_accepts_a_foreign_audio_config(key):
    cls = registry.get_model_class(key)
    kw = {"device": "cpu", "audio_config": _ForeignAudioConfig(representation="raw", normalize=False, target_length_seconds=10)}
    if "return_features_only" in inspect.signature(cls.__init__).parameters:
        kw["return_features_only"] = True
    m = cls(**kw)
    if key != "eat_hf":                                   # EAT replaces the processor by its dedicated frontend (eat_hf.py:185-195)
        assert m.audio_processor.target_length_seconds == 10 and m.audio_processor.sr == 16000


def test_model_base_survives_a_second_base_with_a_required_init():
    """The documented binding lists the reference's ModelBase behind the HIP class; its __init__(device, audio_config) must not be
    reached through super() (it would raise TypeError for the missing arguments)."""
    import torch.nn as nn

    class OtherBase(nn.Module):
        def __init__(self, device, audio_config=None):
            raise AssertionError("second base initialised")

    class Bound(registry.get_model_class("beats"), OtherBase):
        name = "bound_for_test"

    m = Bound(device="cpu", return_features_only=True)
    assert isinstance(m, OtherBase) and len(m.get_model_layers()) == 13


@pytest.mark.skipif(not os.path.isdir("/root/reference/avex"), reason="the reference tree exists in the development container only")
def test_contract_reproduces_against_the_reference_here(golden_dir):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(golden_dir, "make_binding_contract.py"), "--check"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
