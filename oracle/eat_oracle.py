"""CPU oracle: NumPy restatement of the EAT (Data2Vec-multi image encoder) embedding path.

TEST INFRASTRUCTURE ONLY (same rules as beats_oracle.py).

PARITY UNPINNED.  The reference wrapper (avex/models/eat_hf.py:201) obtains its backbone with
``transformers.AutoModel.from_pretrained("worstchan/EAT-base_epoch30_pretrain", trust_remote_code=True)``: the encoder's
arithmetic is HF Hub remote code (revision unpinned), not under /root/reference, not installable here (no network; and the
reference itself notes it is incompatible with transformers >= 5, eat_hf.py:6-21, which is what this image has).  Nothing in
the development container can produce a golden for the encoder.  What the reference tree DOES fix, and this file follows:

  * input: ``spec = process_audio(x)`` -> ``(B, 1024, 128)`` log-mel (EATAudioProcessor, pinned transitively, see
    beats_oracle.eat_preprocess), ``unsqueeze(1)``, ``backbone.extract_features(spec)`` -> ``(B, 513, 768)`` (eat_hf.py:270-278;
    docs/embedding_extraction.md: 512 patches + CLS);
  * parameter names: ``model.`` prefix, ``model.pre_norm`` = fairseq's ``modality_encoders.IMAGE.context_encoder.norm``
    (eat_hf.py:55-73) -- i.e. a post-LN Data2Vec-2.0 encoder whose context-encoder norm runs BEFORE the blocks;
  * hookable layers ``backbone.model.blocks.{i}.attn.proj`` (eat_hf.py:220-236): timm-style blocks with fused qkv.

The rest restates the published EAT / data2vec 2.0 architecture (EAT: Chen et al. 2024, "EAT: Self-Supervised Pre-Training with
Efficient Audio Transformer"; fairseq examples/data2vec ``AltBlock`` with ``layer_norm_first = False``):

  local_encoder   Conv2d(1, 768, 16, stride 16) + bias over the [1024, 128] image -> 64 x 8 = 512 patches, row-major (t, f)
  positions       x += fixed 2-D sin/cos table[:512]   (a buffer in the checkpoint: ``fixed_positional_encoder.positions``)
  class token     x = cat(extra_tokens, x) -> 513 tokens
  pre_norm        LayerNorm(768, eps 1e-6)
  12 x AltBlock   x = x + proj(softmax(q k^T / 8) v);  r = x = norm1(x);  x = norm2(r + fc2(GELU_erf(fc1(x))))
  output          the last block's output (B, 513, 768)

and is the checker for the GPU path on the same synthetic weights.  Shared building blocks (layer_norm, gelu_erf, linear,
attention without bias) are the pinned ones of beats_oracle.py.
"""
from __future__ import annotations

from typing import Dict, Mapping, Tuple

import numpy as np

from . import beats_oracle as O


def patchify(spec: np.ndarray, P: int) -> np.ndarray:
    """``[B, T, F]`` image -> ``[B, (T/P) * (F/P), P*P]`` patch rows, token = t * (F/P) + f, element = dt * P + df (Conv2d order)."""
    B, T, F = spec.shape
    nt, nf = T // P, F // P
    return spec[:, :nt * P, :nf * P].reshape(B, nt, P, nf, P).transpose(0, 1, 3, 2, 4).reshape(B, nt * nf, P * P)


def mhsa(x: np.ndarray, wqkv: np.ndarray, bqkv: np.ndarray, H: int) -> np.ndarray:
    """timm / data2vec ``AltAttention`` core without the output projection: fused qkv Linear, heads, softmax(q k^T * hd^-0.5) v."""
    B, T, E = x.shape
    hd = E // H
    qkv = O.linear(x, wqkv, bqkv).reshape(B, T, 3, H, hd).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q * np.float32(hd ** -0.5)) @ k.transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    a = (e / e.sum(-1, keepdims=True)).astype(np.float32) @ v
    return a.transpose(0, 2, 1, 3).reshape(B, T, E)


def eat_encode(spec: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object], pre: str = "backbone.model."
               ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """``(B, 1024, 128)`` normalised log-mel -> ``(features (B, 513, 768), taps)`` with ``taps[f"backbone.model.blocks.{i}.attn.proj"]``
    = that Linear's raw output (what a forward hook captures)."""
    p = {k[len(pre):]: np.asarray(v, np.float32) for k, v in sd.items() if k.startswith(pre)}
    E = int(cfg["embed_dim"]); L = int(cfg["depth"]); H = int(cfg["num_heads"]); P = int(cfg["patch_size"])
    eps = float(cfg.get("norm_eps", 1e-6))
    x = O.linear(patchify(np.asarray(spec, np.float32), P), p["local_encoder.proj.weight"].reshape(E, P * P), p["local_encoder.proj.bias"])
    n = x.shape[1]
    x = x + p["fixed_positional_encoder.positions"][:, :n]
    x = np.concatenate([np.broadcast_to(p["extra_tokens"], (x.shape[0], 1, E)), x], axis=1).astype(np.float32)
    x = O.layer_norm(x, p["pre_norm.weight"], p["pre_norm.bias"], eps)
    taps: Dict[str, np.ndarray] = {}
    for i in range(L):
        b = f"blocks.{i}."
        a = O.linear(mhsa(x, p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"], H), p[b + "attn.proj.weight"], p[b + "attn.proj.bias"])
        taps[f"backbone.model.blocks.{i}.attn.proj"] = a.copy()
        x = x + a
        r = x = O.layer_norm(x, p[b + "norm1.weight"], p[b + "norm1.bias"], eps)
        h = O.gelu_erf(O.linear(x, p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"]))
        x = O.layer_norm(r + O.linear(h, p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"]), p[b + "norm2.weight"], p[b + "norm2.bias"], eps)
    return x, taps


def eat_forward(wav: np.ndarray, sd: Mapping[str, np.ndarray], cfg: Mapping[str, object], norm_mean: float = -4.268, norm_std: float = 4.569
                ) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """``EATHFModel.forward`` in features mode (eat_hf.py:241-281): waveform -> EATAudioProcessor -> backbone.extract_features."""
    spec = O.eat_preprocess(wav, target_length=int(cfg["target_length"]), n_mels=int(cfg["n_mels"]), norm_mean=norm_mean, norm_std=norm_std)
    return eat_encode(spec, sd, cfg)
