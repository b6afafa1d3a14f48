"""CPU oracle: NumPy restatement of the AVES (wav2vec2-base) embedding path.

TEST INFRASTRUCTURE ONLY (same rules as beats_oracle.py).

PARITY UNPINNED.  The reference wrapper (avex/models/aves_model.py:62-151) builds ``torchaudio.models.wav2vec2_model`` with the
``AVESConfig`` defaults (:19-47) and returns ``self.model.extract_features(x)[0][-1]`` (:149-150).  torchaudio (2.11.0 in the
reference's uv.lock) is third-party, is not under /root/reference and is not installed here, and the AVES weights are behind a URL
(:87-90): nothing in the development container can produce a golden for this path.  This file restates torchaudio's published
wav2vec2 architecture from its documentation as the reference configures it:

  feature extractor  7 x [Conv1d(no bias) -> (layer 0 only) GroupNorm(512, 512) -> GELU], kernels/strides (10,5) (3,2)x4 (2,2)x2
  feature projection LayerNorm(512) -> Linear(512, 768)                                  (dropout: eval identity)
  transformer        x = x + GELU(pos_conv(x)) (weight-normed Conv1d(768, 768, 128, padding 64, groups 16), last frame dropped);
                     x = LayerNorm(x) (layer_norm_first False); 12 post-LN layers:
                     x = LN(x + out_proj(SDPA(q, k, v)));  x = final_LN(x + output_dense(GELU(intermediate_dense(x))))
  output             the last layer's output, [B, 499, 768] for 10 s at 16 kHz

and is the checker for the GPU path on the same synthetic weights.  Second opinion: tests/test_independent_pins.py compares this
file with transformers' Wav2Vec2Model (torchaudio's import_huggingface_model key map) on the same weights: < 2e-5 rel-L2 at 12 layers.  Shared building blocks (layer_norm, gelu_erf, linear,
pos_conv, attention without bias) are the pinned ones of beats_oracle.py.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Tuple

import numpy as np

from . import beats_oracle as O


def conv1d(x: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """``x [B, Cin, T]``, ``w [Cout, Cin, k]`` -> ``[B, Cout, (T-k)//stride+1]`` (no padding, no bias), fp32."""
    B, Cin, T = x.shape
    Cout, _, k = w.shape
    F = (T - k) // stride + 1
    idx = np.arange(F)[:, None] * stride + np.arange(k)[None, :]
    win = x[:, :, idx]                                            # [B, Cin, F, k]
    return np.einsum("bcfk,ock->bof", win, w, optimize=True).astype(np.float32)


def feature_extractor(wav: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object], pre: str = "model.") -> np.ndarray:
    """``[B, T]`` -> ``[B, frames, 512]`` (aves_model.py:25-33: conv layer config; group_norm mode)."""
    x = np.asarray(wav, np.float32)[:, None, :]
    for i, (_c, _k, s) in enumerate(cfg["extractor_conv_layer_config"]):
        x = conv1d(x, sd[pre + f"feature_extractor.conv_layers.{i}.conv.weight"], int(s))
        if i == 0:
            mu = x.mean(axis=2, keepdims=True, dtype=np.float32)
            var = ((x - mu) ** 2).mean(axis=2, keepdims=True, dtype=np.float32)
            x = (x - mu) / np.sqrt(var + np.float32(1e-5))
            x = x * sd[pre + "feature_extractor.conv_layers.0.layer_norm.weight"][None, :, None] + \
                sd[pre + "feature_extractor.conv_layers.0.layer_norm.bias"][None, :, None]
        x = O.gelu_erf(x.astype(np.float32))
    return np.ascontiguousarray(x.transpose(0, 2, 1))


def aves_forward(wav: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object], pre: str = "model."
                 ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """Returns (last layer output ``[B, T', 768]``, taps: ``{layer name: output_dense output [B, T', 768]}``)."""
    E = int(cfg["encoder_embed_dim"]); H = int(cfg["encoder_num_heads"]); L = int(cfg["encoder_num_layers"])
    G = int(cfg["encoder_pos_conv_groups"])
    x = feature_extractor(wav, sd, cfg, pre)
    e = pre + "encoder."
    x = O.layer_norm(x, sd[e + "feature_projection.layer_norm.weight"], sd[e + "feature_projection.layer_norm.bias"])
    x = O.linear(x, sd[e + "feature_projection.projection.weight"], sd[e + "feature_projection.projection.bias"])
    t = e + "transformer."
    w = O.pos_conv_weight(sd[t + "pos_conv_embed.conv.parametrizations.weight.original0"],
                          sd[t + "pos_conv_embed.conv.parametrizations.weight.original1"])
    x = x + O.pos_conv(x, w, sd[t + "pos_conv_embed.conv.bias"], G)
    x = O.layer_norm(x, sd[t + "layer_norm.weight"], sd[t + "layer_norm.bias"])
    taps: Dict[str, np.ndarray] = {}
    B, T, _ = x.shape
    hd = E // H
    for i in range(L):
        p = t + f"layers.{i}."
        q = O.linear(x, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
        k = O.linear(x, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
        v = O.linear(x, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"]).reshape(B, T, H, hd).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) * np.float32(hd ** -0.5)
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s); pr = pr / pr.sum(-1, keepdims=True)
        a = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, E).astype(np.float32)
        a = O.linear(a, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        x = O.layer_norm(x + a, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
        h = O.gelu_erf(O.linear(x, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"]))
        y = O.linear(h, sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
        taps[(p + "feed_forward.output_dense")] = y
        x = O.layer_norm(x + y, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])
    return x, taps
