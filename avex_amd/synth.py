"""Deterministic synthetic weights and clips (no dependence on torch/numpy RNG streams).

There is no network on either box, so neither the official checkpoints
(``hf://EarthSpeciesProject/...``) nor real audio are reachable.  Everything that
needs "a BEATs checkpoint" (bench, smoke, parity tests, golden generation) uses the
closed-form generator below: a counter-based PRNG (splitmix64 keyed by tensor name
and flat element index, Box-Muller to N(0,1)) scaled per tensor to the reference's
own init statistics (reference init code:
``avex/models/beats/backbone.py:59-62,109-122,577-600``).  The same generator runs
in this container (to load the reference for goldens) and on the GPU box.

State-dict keys follow the reference wrapper exactly (``backbone.`` prefix,
``avex/models/beats_model.py:160,189`` and SURVEY.md §8 a13).
"""
from __future__ import annotations

import math
from typing import Dict, Mapping

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _uniform01(key: int, n: int, stream: int) -> np.ndarray:
    """n doubles in (0,1): element i depends only on (key, stream, i)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(key) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        bits = _splitmix64(_splitmix64(idx + base) ^ np.uint64(stream + 1))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name: str, shape, std: float = 1.0, seed: int = 0) -> np.ndarray:
    """N(0, std^2) float32 tensor keyed by (name, seed, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = _fnv1a64(f"{seed}:{name}")
    u1 = _uniform01(key, n, 0)
    u2 = _uniform01(key, n, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def uniform(name: str, shape, bound: float, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    key = _fnv1a64(f"{seed}:{name}")
    u = _uniform01(key, n, 0)
    return ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------
# BEATs configuration (defaults = reference BEATsConfig, avex/models/beats/beats.py:166-228)
# ----------------------------------------------------------------------------------
BEATS_BASE_CFG: Dict[str, object] = dict(
    input_patch_size=16, embed_dim=512, conv_bias=False,
    encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12, activation_fn="gelu",
    layer_norm_first=False, deep_norm=True,
    conv_pos=128, conv_pos_groups=16,
    relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True,
    sample_frequency=16000.0, num_mel_bins=128, frame_length=25.0, frame_shift=10.0,
    fbank_mean=15.41663, fbank_std=6.55582,
    finetuned_model=True, predictor_class=527,
)

# tiny config used for per-stage goldens (verified to run in the reference)
BEATS_TINY_CFG: Dict[str, object] = dict(
    BEATS_BASE_CFG,
    encoder_layers=2, encoder_embed_dim=64, encoder_ffn_embed_dim=128,
    encoder_attention_heads=4, embed_dim=32, conv_pos=16, conv_pos_groups=4,
    num_buckets=32, max_distance=64, finetuned_model=False,
)


# small config on the kernels' geometry (head_dim 64, positional conv k = 128 with 48 channels per group, tile-multiple widths): the base of
# the configuration-space variants below (BEATsConfig options no official checkpoint uses, beats.py:181-212)
BEATS_SMALL_CFG: Dict[str, object] = dict(
    BEATS_BASE_CFG,
    encoder_layers=2, encoder_embed_dim=384, encoder_ffn_embed_dim=768, encoder_attention_heads=6, embed_dim=256,
    conv_pos=128, conv_pos_groups=8, finetuned_model=False,
)
BEATS_VARIANTS: Dict[str, Dict[str, object]] = {
    # pre-LN blocks (backbone.py:328-348, final LayerNorm after the stack :146-147), ReLU, patch-embedding bias
    "preln_relu_convbias": dict(BEATS_SMALL_CFG, layer_norm_first=True, deep_norm=False, activation_fn="relu", conv_bias=True),
    # post-LN without DeepNorm (alpha = 1), tanh-form GELU (modules.py:177-188), no relative position bias at all
    "postln_geluacc_norel": dict(BEATS_SMALL_CFG, deep_norm=False, activation_fn="gelu_accurate", relative_position_embedding=False),
    # post-LN with the gated-linear-unit FFN (fc1 = GLU_Linear(E, F, "swish"), backbone.py:296-297) and an UNGATED position bias
    # (the reference cannot build glu with deep_norm: its DeepNorm init reads fc1.weight, backbone.py:121)
    "postln_glu_nogate": dict(BEATS_SMALL_CFG, deep_norm=False, activation_fn="glu", gru_rel_pos=False),
    # pre-LN with tanh, encoder width = patch width (no post_extract_proj, beats.py:357-358)
    "preln_tanh_nopost": dict(BEATS_SMALL_CFG, layer_norm_first=True, deep_norm=False, activation_fn="tanh", embed_dim=384),
    # "linear" FFN (identity activation)
    "postln_linear": dict(BEATS_SMALL_CFG, activation_fn="linear"),
}


def beats_state_dict(cfg: Mapping[str, object] = BEATS_BASE_CFG, seed: int = 0,
                     include_predictor: bool = True, nontrivial_affine: bool = True
                     ) -> Dict[str, np.ndarray]:
    """Synthetic BEATs wrapper state dict (fp32 numpy), reference key names.

    Scales follow the reference's init.  ``nontrivial_affine``
    perturbs LayerNorm weight/bias, Linear biases and ``grep_a`` away from their
    1/0 init so that parity tests exercise every term (a trained checkpoint has
    non-trivial values there).
    """
    E = int(cfg["encoder_embed_dim"]); F = int(cfg["encoder_ffn_embed_dim"])
    H = int(cfg["encoder_attention_heads"]); L = int(cfg["encoder_layers"])
    D = int(cfg["embed_dim"]); P = int(cfg["input_patch_size"])
    KP = int(cfg["conv_pos"]); G = int(cfg["conv_pos_groups"])
    NB = int(cfg["num_buckets"]); hd = E // H
    beta = (8.0 * L) ** -0.25
    sd: Dict[str, np.ndarray] = {}
    pre = "backbone."

    def n(name, shape, std):
        sd[pre + name] = normal(name, shape, std, seed)

    def aff(name, dim, is_weight):
        if nontrivial_affine:
            v = normal(name, (dim,), 0.1 if is_weight else 0.05, seed)
            sd[pre + name] = (v + 1.0).astype(np.float32) if is_weight else v
        else:
            sd[pre + name] = (np.ones if is_weight else np.zeros)((dim,), np.float32)

    def bias(name, dim, std=0.02):
        sd[pre + name] = normal(name, (dim,), std, seed) if nontrivial_affine else np.zeros((dim,), np.float32)

    if D != E:
        n("post_extract_proj.weight", (E, D), 1.0 / math.sqrt(3.0 * D))      # U(+-1/sqrt(D)) std
        sd[pre + "post_extract_proj.bias"] = uniform("post_extract_proj.bias", (E,), 1.0 / math.sqrt(D), seed)
    sd[pre + "patch_embedding.weight"] = uniform("patch_embedding.weight", (D, 1, P, P), 1.0 / P, seed)
    if bool(cfg.get("conv_bias", False)):
        sd[pre + "patch_embedding.bias"] = uniform("patch_embedding.bias", (D,), 1.0 / P, seed)
    aff("layer_norm.weight", D, True); aff("layer_norm.bias", D, False)

    std_pc = math.sqrt(4.0 / (KP * E))
    v = normal("encoder.pos_conv.0.parametrizations.weight.original1", (E, E // G, KP), std_pc, seed)
    sd[pre + "encoder.pos_conv.0.parametrizations.weight.original1"] = v
    g = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    if nontrivial_affine:
        g = (g * (1.0 + normal("encoder.pos_conv.0.parametrizations.weight.original0", (1, 1, KP), 0.1, seed))).astype(np.float32)
    sd[pre + "encoder.pos_conv.0.parametrizations.weight.original0"] = g
    bias("encoder.pos_conv.0.bias", E)
    aff("encoder.layer_norm.weight", E, True); aff("encoder.layer_norm.bias", E, False)

    xav = math.sqrt(2.0 / (E + E))
    rel = normal("encoder.layers.0.self_attn.relative_attention_bias.weight", (NB, H), 0.02 if not nontrivial_affine else 0.3, seed)
    has_rel = bool(cfg.get("relative_position_embedding", True))
    has_gate = bool(cfg.get("gru_rel_pos", True))
    glu = str(cfg.get("activation_fn", "gelu")) == "glu"
    for i in range(L):
        p = f"encoder.layers.{i}."
        n(p + "self_attn.q_proj.weight", (E, E), xav); bias(p + "self_attn.q_proj.bias", E)
        n(p + "self_attn.k_proj.weight", (E, E), xav); bias(p + "self_attn.k_proj.bias", E)
        n(p + "self_attn.v_proj.weight", (E, E), xav * beta); bias(p + "self_attn.v_proj.bias", E)
        n(p + "self_attn.out_proj.weight", (E, E), xav * beta); bias(p + "self_attn.out_proj.bias", E)
        if has_gate:
            n(p + "self_attn.grep_linear.weight", (8, hd), 0.02 if not nontrivial_affine else 0.1)
            bias(p + "self_attn.grep_linear.bias", 8, 0.1)
            ga = np.ones((1, H, 1, 1), np.float32)
            if nontrivial_affine:
                ga = (ga + normal(p + "self_attn.grep_a", (1, H, 1, 1), 0.2, seed)).astype(np.float32)
            sd[pre + p + "self_attn.grep_a"] = ga
        if has_rel:
            # the table is one shared Parameter (backbone.py:100-103); state_dict lists it per layer
            sd[pre + p + "self_attn.relative_attention_bias.weight"] = rel
        aff(p + "self_attn_layer_norm.weight", E, True); aff(p + "self_attn_layer_norm.bias", E, False)
        if glu:      # GLU_Linear: one Linear(E, 2F); value half first, gate half second (modules.py:150-171)
            n(p + "fc1.linear.weight", (2 * F, E), math.sqrt(2.0 / (E + F)) * beta); bias(p + "fc1.linear.bias", 2 * F)
        else:
            n(p + "fc1.weight", (F, E), math.sqrt(2.0 / (E + F)) * beta); bias(p + "fc1.bias", F)
        n(p + "fc2.weight", (E, F), math.sqrt(2.0 / (E + F)) * beta); bias(p + "fc2.bias", E)
        aff(p + "final_layer_norm.weight", E, True); aff(p + "final_layer_norm.bias", E, False)
    if include_predictor and bool(cfg.get("finetuned_model", False)):
        n("predictor.weight", (int(cfg["predictor_class"]), E), 0.02)
        sd[pre + "predictor.bias"] = np.zeros((int(cfg["predictor_class"]),), np.float32)
    return sd


AVES_BASE_CFG: Dict[str, object] = dict(        # avex/models/aves_model.py:19-47 (AVESConfig defaults = wav2vec2 base)
    extractor_conv_layer_config=[[512, 10, 5], [512, 3, 2], [512, 3, 2], [512, 3, 2], [512, 3, 2], [512, 2, 2], [512, 2, 2]],
    encoder_embed_dim=768, encoder_pos_conv_kernel=128, encoder_pos_conv_groups=16, encoder_num_layers=12,
    encoder_num_heads=12, encoder_ff_interm_features=3072, encoder_layer_norm_first=False,
)


def aves_state_dict(cfg: Mapping[str, object] = AVES_BASE_CFG, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic AVES / wav2vec2-base state dict (fp32 numpy) with torchaudio's key names under the wrapper's ``model.`` prefix
    (aves_model.py:86): conv feature extractor (no bias, GroupNorm on layer 0), feature projection, weight-normed positional
    conv, post-LN transformer.  Scales keep activations O(1) through the stack; LayerNorm / GroupNorm affines and biases are
    perturbed away from 1 / 0 so every term is exercised."""
    E = int(cfg["encoder_embed_dim"]); F = int(cfg["encoder_ff_interm_features"]); L = int(cfg["encoder_num_layers"])
    KP = int(cfg["encoder_pos_conv_kernel"]); G = int(cfg["encoder_pos_conv_groups"])
    convs = [tuple(int(v) for v in c) for c in cfg["extractor_conv_layer_config"]]
    sd: Dict[str, np.ndarray] = {}
    pre = "model."

    def n(name, shape, std):
        sd[pre + name] = normal("aves." + name, shape, std, seed)

    def aff(name, dim, is_weight):
        v = normal("aves." + name, (dim,), 0.1 if is_weight else 0.05, seed)
        sd[pre + name] = (v + 1.0).astype(np.float32) if is_weight else v

    cin = 1
    for i, (cout, k, _s) in enumerate(convs):
        n(f"feature_extractor.conv_layers.{i}.conv.weight", (cout, cin, k), math.sqrt(2.0 / (cin * k)))   # kaiming_normal
        cin = cout
    aff("feature_extractor.conv_layers.0.layer_norm.weight", convs[0][0], True)
    aff("feature_extractor.conv_layers.0.layer_norm.bias", convs[0][0], False)
    C = convs[-1][0]
    aff("encoder.feature_projection.layer_norm.weight", C, True); aff("encoder.feature_projection.layer_norm.bias", C, False)
    n("encoder.feature_projection.projection.weight", (E, C), 1.0 / math.sqrt(C)); n("encoder.feature_projection.projection.bias", (E,), 0.02)
    v = normal("aves.pos_conv.v", (E, E // G, KP), math.sqrt(4.0 / (KP * E)), seed)
    g = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    g = (g * (1.0 + normal("aves.pos_conv.g", (1, 1, KP), 0.1, seed))).astype(np.float32)
    sd[pre + "encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original0"] = g
    sd[pre + "encoder.transformer.pos_conv_embed.conv.parametrizations.weight.original1"] = v
    n("encoder.transformer.pos_conv_embed.conv.bias", (E,), 0.02)
    aff("encoder.transformer.layer_norm.weight", E, True); aff("encoder.transformer.layer_norm.bias", E, False)
    for i in range(L):
        p = f"encoder.transformer.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            n(p + f"attention.{nm}.weight", (E, E), math.sqrt(1.0 / E) * (1.0 if nm in ("q_proj", "k_proj") else 0.5))
            n(p + f"attention.{nm}.bias", (E,), 0.02)
        aff(p + "layer_norm.weight", E, True); aff(p + "layer_norm.bias", E, False)
        n(p + "feed_forward.intermediate_dense.weight", (F, E), math.sqrt(2.0 / (E + F)) * 0.7); n(p + "feed_forward.intermediate_dense.bias", (F,), 0.02)
        n(p + "feed_forward.output_dense.weight", (E, F), math.sqrt(2.0 / (E + F)) * 0.7); n(p + "feed_forward.output_dense.bias", (E,), 0.02)
        aff(p + "final_layer_norm.weight", E, True); aff(p + "final_layer_norm.bias", E, False)
    return sd


EAT_BASE_CFG: Dict[str, object] = dict(         # EAT-base (Data2Vec-multi image encoder on a [1024, 128] mel image); parity unpinned
    embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, patch_size=16, target_length=1024, n_mels=128, max_length=768,
    norm_eps=1e-6, layer_norm_first=False,
)


def sincos_2d_positions(embed_dim: int, grid_h: int, grid_w: int) -> np.ndarray:
    """MAE-style fixed 2-D sine/cosine position table ``[grid_h * grid_w, embed_dim]`` (first half of the channels encodes the
    column index w, second half the row index h; each half is [sin | cos] over ``omega_k = 10000^(-k / (D/4))``), row-major
    over (h, w) like the patch order -- what EAT's ``fixed_positional_encoder.positions`` buffer holds."""
    def one_d(d: int, pos: np.ndarray) -> np.ndarray:
        omega = 1.0 / 10000 ** (np.arange(d // 2, dtype=np.float64) / (d / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh, gw = np.arange(grid_h, dtype=np.float32), np.arange(grid_w, dtype=np.float32)
    grid = np.meshgrid(gw, gh)                               # [0] = w index, [1] = h index, each [grid_h, grid_w]
    emb_h = one_d(embed_dim // 2, grid[0])
    emb_w = one_d(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1).astype(np.float32)


def eat_state_dict(cfg: Mapping[str, object] = EAT_BASE_CFG, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic EAT-base state dict (fp32 numpy) under the avex wrapper's ``backbone.`` prefix with the HF remote model's key
    names (``model.local_encoder.proj``, ``model.extra_tokens``, ``model.fixed_positional_encoder.positions``, ``model.pre_norm``,
    ``model.blocks.{i}.{attn.qkv, attn.proj, norm1, mlp.fc1, mlp.fc2, norm2}``; the fairseq -> HF renaming the reference does,
    avex/models/eat_hf.py:55-73, fixes ``model.`` + ``pre_norm``; ``blocks.{i}.attn.proj`` is the hook name, :220-236)."""
    E = int(cfg["embed_dim"]); L = int(cfg["depth"]); F = E * int(cfg["mlp_ratio"]); P = int(cfg["patch_size"])
    gw = int(cfg["n_mels"]) // P
    sd: Dict[str, np.ndarray] = {}
    pre = "backbone.model."

    def n(name, shape, std):
        sd[pre + name] = normal("eat." + name, shape, std, seed)

    def aff(name, dim, is_weight):
        v = normal("eat." + name, (dim,), 0.1 if is_weight else 0.05, seed)
        sd[pre + name] = (v + 1.0).astype(np.float32) if is_weight else v

    n("local_encoder.proj.weight", (E, 1, P, P), 1.0 / P); n("local_encoder.proj.bias", (E,), 0.02)
    n("extra_tokens", (1, 1, E), 0.02)
    sd[pre + "fixed_positional_encoder.positions"] = sincos_2d_positions(E, int(cfg["max_length"]), gw)[None]
    aff("pre_norm.weight", E, True); aff("pre_norm.bias", E, False)
    for i in range(L):
        p = f"blocks.{i}."
        n(p + "attn.qkv.weight", (3 * E, E), math.sqrt(1.0 / E) * 0.8); n(p + "attn.qkv.bias", (3 * E,), 0.02)
        n(p + "attn.proj.weight", (E, E), math.sqrt(1.0 / E) * 0.5); n(p + "attn.proj.bias", (E,), 0.02)
        aff(p + "norm1.weight", E, True); aff(p + "norm1.bias", E, False)
        n(p + "mlp.fc1.weight", (F, E), math.sqrt(2.0 / (E + F)) * 0.7); n(p + "mlp.fc1.bias", (F,), 0.02)
        n(p + "mlp.fc2.weight", (E, F), math.sqrt(2.0 / (E + F)) * 0.7); n(p + "mlp.fc2.bias", (E,), 0.02)
        aff(p + "norm2.weight", E, True); aff(p + "norm2.bias", E, False)
    return sd


# torchvision efficientnet_b0: (expand ratio, kernel, stride, in, out, layers) per stage
EFFNET_B0_STAGES = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
                    (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]


# torchvision efficientnet_b1 = width multiplier 1.0, depth multiplier 1.1: the same channel widths, ceil(1.1 n) layers per stage
EFFNET_B1_STAGES = [(er, k, st, cin, cout, -(-(11 * n) // 10)) for (er, k, st, cin, cout, n) in EFFNET_B0_STAGES]
EFFNET_STAGES = {"b0": EFFNET_B0_STAGES, "b1": EFFNET_B1_STAGES}


def effnet_b0_state_dict(seed: int = 0, stages=EFFNET_B0_STAGES, head: int = 1280) -> Dict[str, np.ndarray]:
    """Synthetic EfficientNet-B0 ``features`` state dict (fp32 numpy) with torchvision's key names under the wrapper's ``model.``
    prefix (efficientnet.py:57): Conv2dNormActivation = ``.0`` conv (no bias) + ``.1`` BatchNorm2d; MBConv ``block`` =
    [expand], depthwise, SqueezeExcitation (fc1 / fc2 1x1 convs with bias, squeeze = in // 4), project.  BatchNorm running
    statistics and affines are non-trivial so the folding is exercised."""
    sd: Dict[str, np.ndarray] = {}
    pre = "model."

    def conv(name, cout, cin, k):
        sd[pre + name + ".weight"] = normal("effnet." + name, (cout, cin, k, k), math.sqrt(2.0 / (cin * k * k)), seed)

    def bn(name, c):
        sd[pre + name + ".weight"] = (1.0 + normal("effnet." + name + ".w", (c,), 0.1, seed)).astype(np.float32)
        sd[pre + name + ".bias"] = normal("effnet." + name + ".b", (c,), 0.1, seed)
        sd[pre + name + ".running_mean"] = normal("effnet." + name + ".m", (c,), 0.1, seed)
        sd[pre + name + ".running_var"] = (1.0 + 0.3 * np.abs(normal("effnet." + name + ".v", (c,), 1.0, seed))).astype(np.float32)
        sd[pre + name + ".num_batches_tracked"] = np.zeros((), np.int64)

    conv("features.0.0", stages[0][3], 3, 3); bn("features.0.1", stages[0][3])
    for si, (er, k, _s, cin, cout, n) in enumerate(stages, start=1):
        for j in range(n):
            ci = cin if j == 0 else cout
            ce = ci * er
            p = f"features.{si}.{j}.block."
            d = 0
            if er != 1:
                conv(p + "0.0", ce, ci, 1); bn(p + "0.1", ce); d = 1
            sd[pre + p + f"{d}.0.weight"] = normal("effnet." + p + "dw", (ce, 1, k, k), math.sqrt(2.0 / (k * k)), seed)
            bn(p + f"{d}.1", ce)
            cs = max(1, ci // 4)
            sd[pre + p + f"{d + 1}.fc1.weight"] = normal("effnet." + p + "fc1", (cs, ce, 1, 1), math.sqrt(1.0 / ce), seed)
            sd[pre + p + f"{d + 1}.fc1.bias"] = normal("effnet." + p + "fc1b", (cs,), 0.1, seed)
            sd[pre + p + f"{d + 1}.fc2.weight"] = normal("effnet." + p + "fc2", (ce, cs, 1, 1), math.sqrt(1.0 / cs), seed)
            sd[pre + p + f"{d + 1}.fc2.bias"] = normal("effnet." + p + "fc2b", (ce,), 0.5, seed)
            conv(p + f"{d + 2}.0", cout, ce, 1); bn(p + f"{d + 2}.1", cout)
    last = len(stages) + 1
    conv(f"features.{last}.0", head, stages[-1][4], 1); bn(f"features.{last}.1", head)
    return sd


# ----------------------------------------------------------------------------------
# Synthetic clips (BASELINE.md §3 "Inputs")
# ----------------------------------------------------------------------------------
def noise_clips(batch: int, samples: int, seed: int = 0, first_clip: int = 0, amp: float = 0.1) -> np.ndarray:
    """``amp * N(0,1)`` clips keyed by global clip index (shards are reproducible)."""
    out = np.empty((batch, samples), np.float32)
    for b in range(batch):
        out[b] = normal(f"clip{first_clip + b}", (samples,), amp, seed)
    return out


def tone_clips(samples: int, sr: int = 16000) -> np.ndarray:
    """6 deterministic sines 220/440/880 Hz x amp 0.8/0.9 (mirrors the reference's
    regression inputs, tests/integration/test_official_models_output_regression.py:135-156)."""
    t = np.arange(samples, dtype=np.float64) / sr
    rows = []
    for f in (220.0, 440.0, 880.0):
        for a in (0.8, 0.9):
            rows.append((a * np.sin(2 * math.pi * f * t)).astype(np.float32))
    return np.stack(rows)
