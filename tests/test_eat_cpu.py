"""EAT: what can be checked without a GPU and without the HF remote code.

* the model class mirror: constructor contract and errors, parameter names, hook layers, fairseq key renaming (eat_hf.py:55-73,
  :175-176, :220-236);
* the oracle's arithmetic against an independent restatement made of PyTorch's own ops (conv2d, layer_norm,
  scaled_dot_product_attention, gelu) -- this pins the NumPy code to the documented architecture, NOT to the remote model
  (parity stays unpinned, oracle/eat_oracle.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import avex_amd
from avex_amd import synth
from avex_amd.eat_hf import EATHFModel, rename_fairseq_key
from oracle import eat_oracle as EO


def test_class_contract_cpu():
    assert {"eat_hf", "eathf"} <= set(avex_amd.list_model_classes())
    with pytest.raises(ValueError):
        EATHFModel(device="cpu")                                                   # num_classes required unless features mode
    m = EATHFModel(device="cpu", return_features_only=True)
    assert m.classifier is None and m.audio_processor.target_length == 1024 and m.audio_processor.norm_mean == -4.268
    names = m.get_model_layers()
    assert names == [f"backbone.model.blocks.{i}.attn.proj" for i in range(12)]
    assert m.register_hooks_for_layers([0, -1]) == [names[0], names[-1]]
    assert m.register_hooks_for_layers(["last_layer"]) == [names[-1]]
    with pytest.raises(ValueError):
        m.register_hooks_for_layers(["backbone.model.blocks.99.attn.proj"])
    with pytest.raises(TypeError):
        m.register_hooks_for_layers([True])
    m.deregister_all_hooks()
    with pytest.raises(ValueError):
        m._hook_layers = []
        m.extract_embeddings(torch.zeros(1, 16000))                                # no hooks
    sd = synth.eat_state_dict()
    assert set(sd) == set(m.state_dict())                                          # HF remote model's names under backbone.
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    c = EATHFModel(device="cpu", num_classes=7)
    assert c.classifier.weight.shape == (7, 768) and "classifier.weight" in c.state_dict()
    # every official EAT id builds through the registry in features mode
    for mid in ("esp_aves2_eat_all", "esp_aves2_eat_bio", "esp_aves2_sl_eat_all_ssl_all", "esp_aves2_sl_eat_bio_ssl_all"):
        spec = avex_amd.get_model_spec(mid).model_copy(deep=True)
        mm = avex_amd.build_model_from_spec(spec, "cpu", return_features_only=True)
        assert isinstance(mm, EATHFModel) and mm.norm_mean == -4.268            # eat_norm_mean of the YAML never reaches the class
    info = avex_amd.list_model_layers("esp_aves2_eat_all")
    assert info["last_layer"] == names[-1] and len(info["layers"]) == 12


def test_fairseq_key_renaming_and_loading(tmp_path):
    assert rename_fairseq_key("modality_encoders.IMAGE.context_encoder.norm.weight") == "model.pre_norm.weight"
    assert rename_fairseq_key("modality_encoders.IMAGE.context_encoder.norm.bias") == "model.pre_norm.bias"
    assert rename_fairseq_key("modality_encoders.IMAGE.local_encoder.proj.weight") == "model.local_encoder.proj.weight"
    assert rename_fairseq_key("blocks.3.attn.qkv.weight") == "model.blocks.3.attn.qkv.weight"
    assert rename_fairseq_key("model.blocks.3.attn.qkv.weight") == "model.blocks.3.attn.qkv.weight"
    sd = synth.eat_state_dict(dict(synth.EAT_BASE_CFG, depth=1))
    fs = {}
    for k, v in sd.items():
        k = k[len("backbone.model."):]
        if k.startswith("pre_norm."):
            k = "modality_encoders.IMAGE.context_encoder.norm." + k.split(".")[1]
        elif not k.startswith("blocks."):
            k = "modality_encoders.IMAGE." + k
        fs[k] = torch.from_numpy(v)
    fs["_ema.blocks.0.attn.qkv.weight"] = torch.zeros(1)
    path = tmp_path / "eat_fairseq.pt"
    torch.save({"model": fs}, path)
    m = EATHFModel(device="cpu", return_features_only=True, init_config={"depth": 1}, fairseq_weights_path=str(path))
    got = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], torch.from_numpy(v)), k


def _torch_encode(spec, sd, cfg):
    p = {k[len("backbone.model."):]: torch.from_numpy(v) for k, v in sd.items()}
    E, L, H, eps = int(cfg["embed_dim"]), int(cfg["depth"]), int(cfg["num_heads"]), float(cfg["norm_eps"])
    x = F.conv2d(torch.from_numpy(spec)[:, None], p["local_encoder.proj.weight"], p["local_encoder.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2)                                               # [B, 64 * 8, 768], (t, f) row-major
    x = x + p["fixed_positional_encoder.positions"][:, :x.shape[1]]
    x = torch.cat([p["extra_tokens"].expand(x.shape[0], -1, -1), x], 1)
    x = F.layer_norm(x, (E,), p["pre_norm.weight"], p["pre_norm.bias"], eps)
    taps = {}
    for i in range(L):
        b = f"blocks.{i}."
        B, T, _ = x.shape
        qkv = F.linear(x, p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"]).reshape(B, T, 3, H, E // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, T, E)
        a = F.linear(a, p[b + "attn.proj.weight"], p[b + "attn.proj.bias"])
        taps[i] = a
        x = x + a
        r = x = F.layer_norm(x, (E,), p[b + "norm1.weight"], p[b + "norm1.bias"], eps)
        h = F.linear(F.gelu(F.linear(x, p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"])), p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"])
        x = F.layer_norm(r + h, (E,), p[b + "norm2.weight"], p[b + "norm2.bias"], eps)
    return x, taps


def test_oracle_matches_torch_ops():
    cfg = dict(synth.EAT_BASE_CFG, depth=2)
    sd = synth.eat_state_dict(cfg)
    spec = (synth.normal("eatspec", (2, 1024, 128), 0.5)).astype(np.float32)
    f, taps = EO.eat_encode(spec, sd, cfg)
    with torch.no_grad():
        ft, tt = _torch_encode(spec, sd, cfg)
    assert f.shape == (2, 513, 768)
    assert np.linalg.norm(f - ft.numpy()) / np.linalg.norm(ft.numpy()) < 5e-6
    for i in range(2):
        a = taps[f"backbone.model.blocks.{i}.attn.proj"]
        assert np.linalg.norm(a - tt[i].numpy()) / np.linalg.norm(tt[i].numpy()) < 5e-6


def test_position_table_known_values():
    pos = synth.sincos_2d_positions(768, 768, 8)
    assert pos.shape == (6144, 768)
    assert np.all(pos[0, :192] == 0) and np.all(pos[0, 192:384] == 1)             # (h, w) = (0, 0): sin 0, cos 0
    # token 9 = (h 1, w 1): both halves encode index 1
    assert np.allclose(pos[9, 0], np.sin(1.0), atol=1e-6) and np.allclose(pos[9, 384], np.sin(1.0), atol=1e-6)
    assert np.allclose(pos[8, :384], pos[0, :384]) and not np.allclose(pos[8, 384:], pos[0, 384:])   # first half follows w only
