"""CPU restatement (numpy, fp64 accumulation where it is free) of the reference's probe-head forwards -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  Pinned against the reference's own
LinearProbe / MLPProbe / AttentionProbe run in this container (tests/golden/make_probe_goldens.py -> tests/golden/probes.npz;
tests/test_oracle_probes.py).

Follows /root/reference/avex/models/probes/:
  layer_mix        base_probes.py:197-206  (_sum: softmax(layer_weights) or ones; out = out + w * emb)
  linear_probe     linear_probe.py:44-46,58-68 + base_probes.py:299-322 (2-D: flatten trailing dims, project, _sum)
  mlp_probe        mlp_probe.py:51-73,83-91 (Linear -> act -> Dropout(eval: identity) ... -> Linear)
  attention_probe  attention_probe.py:59-86,110-134 (pos enc add, [MHA -> LN(x + attn)] * L, mean over sequence, classifier);
                   nn.MultiheadAttention(batch_first=True): q,k,v = x W_in^T + b_in split in thirds, heads of E/H, softmax(q k^T /
                   sqrt(E/H)) v, out_proj.
"""
import math

import numpy as np


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def layer_mix(taps, layer_weights=None):
    """base_probes.py:197-206, fp32 like the reference (separate multiply and add, list order)."""
    taps = [np.asarray(t, np.float32) for t in taps]
    w = np.ones(len(taps), np.float32) if layer_weights is None else softmax(np.asarray(layer_weights, np.float32)).astype(np.float32)
    out = np.zeros_like(taps[0])
    for t, wl in zip(taps, w):
        out = out + wl * t
    return out


def linear(x, w, b=None):
    y = np.asarray(x, np.float64) @ np.asarray(w, np.float64).T
    return y + b if b is not None else y


def _act(x, name):
    if name == "relu":
        return np.maximum(x, 0)
    if name == "gelu":
        from scipy.special import erf
        return 0.5 * x * (1 + erf(x / math.sqrt(2.0)))
    if name == "tanh":
        return np.tanh(x)
    raise ValueError(name)


def combine_2d(embs, layer_weights=None):
    """base_probes.py:299-322 for equal feature sizes (no projectors): flatten trailing dims, then _sum."""
    if isinstance(embs, (list, tuple)):
        return layer_mix([np.asarray(e).reshape(len(e), -1) for e in embs], layer_weights)
    return np.asarray(embs).reshape(len(embs), -1)


def linear_probe(embs, sd):
    x = combine_2d(embs, sd.get("layer_weights"))
    return linear(x, sd["classifier.weight"], sd["classifier.bias"])


def mlp_probe(embs, sd, activation="relu"):
    x = combine_2d(embs, sd.get("layer_weights")).astype(np.float64)
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("mlp.")})
    for n, i in enumerate(idx):
        x = linear(x, sd[f"mlp.{i}.weight"], sd[f"mlp.{i}.bias"])
        if n + 1 < len(idx):
            x = _act(x, activation)
    return x


def layernorm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def mha(x, in_w, in_b, out_w, out_b, num_heads, key_pad=None):
    B, T, E = x.shape
    hd = E // num_heads
    qkv = linear(x, in_w, in_b)
    q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, T, num_heads, hd).transpose(0, 2, 1, 3) for i in range(3))
    s = (q / math.sqrt(hd)) @ k.transpose(0, 1, 3, 2)
    if key_pad is not None:
        s = np.where(np.asarray(key_pad, bool)[:, None, None, :], -np.inf, s)
    o = (softmax(s) @ v).transpose(0, 2, 1, 3).reshape(B, T, E)
    return linear(o, out_w, out_b)


def attention_probe(embs, sd, num_heads, key_pad=None):
    x = layer_mix(embs, sd.get("layer_weights")) if isinstance(embs, (list, tuple)) else np.asarray(embs, np.float32)
    x = x.astype(np.float64)
    if "pos_encoding" in sd:
        x = x + sd["pos_encoding"][:, : x.shape[1]]
    if key_pad is not None and np.asarray(key_pad).shape[1] != x.shape[1]:
        key_pad = None                                                       # attention_probe.py:124-125
    n_layers = len({k.split(".")[1] for k in sd if k.startswith("attention_layers.")})
    for i in range(n_layers):
        a = mha(x, sd[f"attention_layers.{i}.in_proj_weight"], sd[f"attention_layers.{i}.in_proj_bias"],
                sd[f"attention_layers.{i}.out_proj.weight"], sd[f"attention_layers.{i}.out_proj.bias"], num_heads, key_pad)
        x = layernorm(x + a, sd[f"layer_norms.{i}.weight"], sd[f"layer_norms.{i}.bias"])
    return linear(x.mean(1), sd["classifier.weight"], sd["classifier.bias"])


def transformer_probe(embs, sd, num_heads, key_pad=None):
    """``TransformerProbe.forward`` (transformer_probe.py:97-116): learned mix of the taps, optional positions, ``num_layers`` post-LN
    ``nn.TransformerEncoderLayer`` s (self attention -> LayerNorm(x + attn) -> ReLU feed-forward -> LayerNorm(x + ff)), mean over the
    sequence, classifier.  With a key padding mask PyTorch's encoder takes its nested-tensor path in eval mode and hands back ZEROS at
    the padded positions, which the mean over all T positions then includes -- restated as such (pinned by probes_seq.npz)."""
    x = layer_mix(embs, sd.get("layer_weights")) if isinstance(embs, (list, tuple)) else np.asarray(embs, np.float32)
    x = x.astype(np.float64)
    if "pos_encoding" in sd:
        x = x + sd["pos_encoding"][:, : x.shape[1]]
    if key_pad is not None and np.asarray(key_pad).shape[1] != x.shape[1]:
        key_pad = None                                                       # transformer_probe.py:109-110
    n_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.layers.")})
    for i in range(n_layers):
        p = f"transformer.layers.{i}."
        a = mha(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"], sd[p + "self_attn.out_proj.weight"],
                sd[p + "self_attn.out_proj.bias"], num_heads, key_pad)
        x = layernorm(x + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        f = linear(np.maximum(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"]), 0.0), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = layernorm(x + f, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    if key_pad is not None:
        x = np.where(np.asarray(key_pad, bool)[..., None], 0.0, x)
    return linear(x.mean(1), sd["classifier.weight"], sd["classifier.bias"])


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of one ``nn.LSTM`` layer, batch first, zero initial state; gate order i, f, g, o (torch.nn.LSTM)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xg = x @ w_ih.T.astype(np.float64) + (b_ih.astype(np.float64) + b_hh.astype(np.float64))
    h = np.zeros((B, H)); c = np.zeros((B, H))
    out = np.zeros((B, T, H))
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = xg[:, t] + h @ w_hh.T.astype(np.float64)
        i, f, gg, o = _sigmoid(g[:, :H]), _sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), _sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    return out


def lstm_probe(embs, sd):
    """``LSTMProbe.forward`` (lstm_probe.py:87-104): learned mix, optional positions, ``nn.LSTM`` (layers / directions read off the state
    dict), mean over the sequence, classifier.  (The padding mask does not reach the LSTM in the reference.)"""
    x = layer_mix(embs, sd.get("layer_weights")) if isinstance(embs, (list, tuple)) else np.asarray(embs, np.float32)
    x = x.astype(np.float64)
    if "pos_encoding" in sd:
        x = x + sd["pos_encoding"][:, : x.shape[1]]
    layer = 0
    while f"lstm.weight_ih_l{layer}" in sd:
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            k = f"l{layer}{suffix}"
            if f"lstm.weight_ih_{k}" not in sd:
                continue
            outs.append(lstm_layer(x, sd[f"lstm.weight_ih_{k}"], sd[f"lstm.weight_hh_{k}"], sd[f"lstm.bias_ih_{k}"], sd[f"lstm.bias_hh_{k}"], rev))
        x = np.concatenate(outs, axis=-1)
        layer += 1
    return linear(x.mean(1), sd["classifier.weight"], sd["classifier.bias"])
