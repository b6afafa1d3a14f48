"""Replay, on the GPU and with real forwards, the call sequence the reference's factory / loader / probe factory made on the HIP
classes (`-m gpu`).  The sequence and its arguments come from ``tests/golden/binding_contract.json`` -- recorded by
``tests/golden/make_binding_contract.py`` against the real reference (factory.py:108-166, load.py:152-311,521-570,
probes/base_probes.py:23-195), where the forward itself could not run for want of a GPU.  Here the reference is absent, so its
``AudioConfig`` is stood in for by another pydantic class with the recorded field values, and the probe's two
``extract_embeddings`` calls are made with the recorded shapes and keywords.
"""
import json
import os

import numpy as np
import pytest
import torch
from pydantic import create_model

from avex_amd import registry, synth

pytestmark = pytest.mark.gpu
MIRROR = {"beats_hip": "beats", "eat_hf_hip": "eat_hf", "efficientnet_hip": "efficientnet", "aves_hip": "aves"}
STATE = {"beats_hip": lambda: synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0), "eat_hf_hip": lambda: synth.eat_state_dict(synth.EAT_BASE_CFG, seed=0),
         "efficientnet_hip": lambda: synth.effnet_b0_state_dict(seed=0), "aves_hip": lambda: synth.aves_state_dict(synth.AVES_BASE_CFG, seed=0)}


@pytest.fixture(scope="module")
def contract(golden_dir):
    with open(os.path.join(golden_dir, "binding_contract.json")) as f:
        return json.load(f)


def _foreign_audio_config(fields: dict):
    """An object of a class that is NOT avex_amd.configs.AudioConfig, carrying the recorded values (what factory.py:132-143 passes)."""
    fields = {k: v for k, v in fields.items() if k != "__type__"}
    cls = create_model("AudioConfig", **{k: (type(v) if v is not None else type(None), v) for k, v in fields.items()})
    return cls()


@pytest.mark.parametrize("key", list(MIRROR))
def test_reference_call_sequence_with_real_forwards(built_lib, contract, key):
    fam = contract["families"][key]
    cls = registry.get_model_class(MIRROR[key])
    # ---- the constructor call the reference's factory made (load path), on the GPU
    kw = dict(fam["load"]["init_keywords"])
    kw["audio_config"] = _foreign_audio_config(kw["audio_config"])
    kw["device"] = "cuda"
    m = cls(**kw)
    # ---- _load_checkpoint: load_state_dict(strict=False) of a flat state dict, then .to(device)
    sd = STATE[key]()
    res = m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and len(m.state_dict()) == fam["load"]["model_state_dict_keys"]
    m = m.to("cuda")
    assert m.get_model_layers() == fam["layers"]["layers"]
    # ---- every probe construction the contract recorded
    for name in ("probe_linear_mean", "probe_linear_two_layers", "probe_attention_none"):
        if name not in fam:
            continue
        p = fam[name]
        # build_probe_from_config: register_hooks_for_layers(probe_config.target_layers) (probes/utils/factory.py:146-151)
        assert m.register_hooks_for_layers(p["probe_config"]["target_layers"]) == p["resolved_layers"]
        # _BaseProbe.__init__: freeze (base_probes.py:47-53)
        m.eval()
        for q in m.parameters():
            q.requires_grad = False
        ap = m.audio_processor                                                           # _infer_target_length (:107-119)
        L = int(ap.target_length_seconds * ap.sr) if getattr(ap, "target_length_seconds", None) else int(ap.target_length)
        c = p["construction_calls"][0]
        assert [1, L] == c["input_shape"]
        with torch.no_grad():                                                            # _extract_dummy_embeddings_from_model (:121-125)
            dummy = torch.randn(1, L, device="cuda")
            emb = m.extract_embeddings(dummy, **c["keywords"])
        n = len(p["resolved_layers"])
        width = 5120 if key == "efficientnet_hip" else 768      # (B, 1280, 4, 32) -> mean(-1) -> view(B, 5120) (efficientnet.py:297-311)
        if c["keywords"]["aggregation"] == "none":
            assert isinstance(emb, torch.Tensor) and emb.dim() == 3 and emb.shape[0] == 1 and emb.shape[2] == width
        else:
            assert tuple(emb.shape) == (1, p["inferred_dim"]) == (1, width * n)
        assert emb.is_cuda and emb.dtype == torch.float32 and bool(torch.isfinite(emb).all())
        f = p["forward_calls"][0]                                                        # _get_embeddings (:167-195)
        x = torch.randn(*f["input_shape"], device="cuda") * 0.1
        out = m.extract_embeddings(x, **f["keywords"])
        assert out.shape[0] == 2 and bool(torch.isfinite(out).all())
        # a frozen, detached result is what the probe's head consumes
        assert not out.requires_grad
        assert not m._hook_outputs                                                       # cleared in finally (base_model.py:89-96)
        m.deregister_all_hooks()
