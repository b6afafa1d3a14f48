"""AVES (wav2vec2-base) embedding path on the MI355X kernels.

The reference wrapper (avex/models/aves_model.py:62-151) builds ``torchaudio.models.wav2vec2_model`` with the AVESConfig
defaults and returns the last transformer layer's output.  Everything after the convolutional feature extractor has the shape
of the BEATs encoder this package already runs (LayerNorm(512) -> Linear(512, 768) -> weight-normed grouped positional conv
-> post-LN transformer layers; no relative-position bias, no DeepNorm scale), so it is composed from the same HIP kernels through
the C ABI: ``avexhip_gemm`` (bias / GELU / residual epilogues), ``avexhip_layernorm``, ``avexhip_posconv``,
``avexhip_attention`` (plain softmax), ``avexhip_mean_pool``.  The feature extractor is new:

* layer 0 (one input channel, 10 taps, GroupNorm over time) is ``avexhip_wavconv0`` (two passes over the waveform);
* layers 1-6 are GEMMs on STRIDED ROWS of the previous activations: with ``[clip][frame][channel]`` rows, output frame t of a
  Conv1d(k, s) reads the k * 512 contiguous values starting at frame s * t, i.e. ``lda = s * 512``, ``K = k * 512`` -- no
  im2col buffer.  Per-clip frame counts are padded (P_l = s_{l+1} P_{l+1}) so that one uniform ``lda`` covers the whole batch;
  the padded rows compute values no valid row ever reads.

PARITY UNPINNED: torchaudio is third-party, absent from the reference tree and from both machines, and the AVES checkpoint is
behind a URL; the checker is ``oracle/aves_oracle.py`` on synthetic weights.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _capi
from . import kernels as K

__all__ = ["AvesEncoder", "conv_frame_plan"]


def conv_frame_plan(T: int, convs: Sequence[Sequence[int]]):
    """Valid frames F_l and padded per-clip row counts P_l of every conv layer for T samples (P_{l-1} = stride_l * P_l >= F_{l-1})."""
    F = []
    n = T
    for (_c, k, s) in convs:
        n = (n - k) // s + 1 if n >= k else 0
        F.append(n)
    if F[-1] <= 0:
        raise ValueError(f"audio too short for the feature extractor ({T} samples)")
    pad = 0
    while True:
        P = [0] * len(convs)
        P[-1] = F[-1] + pad
        for l in range(len(convs) - 1, 0, -1):
            P[l - 1] = convs[l][2] * P[l]
        if all(P[l] >= F[l] for l in range(len(convs))):
            return F, P
        pad += 1


class AvesEncoder:
    """``[B, T]`` fp32 waveforms on the GPU -> last-layer features ``[B, T', 768]`` / hook taps / pooled embeddings."""

    def __init__(self, cfg: Mapping[str, object], state: Mapping[str, np.ndarray], operand_dtype: str = "f16", prefix: str = "model.") -> None:
        _capi.require_gpu()
        self.cfg = dict(cfg)
        self.dtype = operand_dtype
        self.convs = [tuple(int(v) for v in c) for c in cfg["extractor_conv_layer_config"]]
        if self.convs[0] != (512, 10, 5) or any(c[0] != 512 for c in self.convs):
            raise K.AvexHipError("AVES feature extractor: only the wav2vec2-base layout (512 channels, first layer k=10 s=5) is built")
        if bool(cfg.get("encoder_layer_norm_first", False)):
            raise K.AvexHipError("AVES: encoder_layer_norm_first=True (pre-LN) is not built")
        self.E = int(cfg["encoder_embed_dim"]); self.H = int(cfg["encoder_num_heads"]); self.L = int(cfg["encoder_num_layers"])
        self.G = int(cfg["encoder_pos_conv_groups"]); self.KP = int(cfg["encoder_pos_conv_kernel"])
        if self.E != 64 * self.H:
            raise K.AvexHipError("AVES: head_dim must be 64")
        dev = torch.device("cuda", torch.cuda.current_device())
        f32 = lambda name: torch.from_numpy(np.ascontiguousarray(np.asarray(state[prefix + name], np.float32))).to(dev)
        half = lambda t: K.to_half(t.contiguous(), operand_dtype)
        fe = "feature_extractor.conv_layers."
        self.w0 = f32(fe + "0.conv.weight").reshape(512, 10)
        self.gn_w, self.gn_b = f32(fe + "0.layer_norm.weight"), f32(fe + "0.layer_norm.bias")
        # conv weights [out, in, k] -> [out, k, in]: the K order of a strided activation row is (frame, channel)
        self.wc = [half(f32(fe + f"{i}.conv.weight").permute(0, 2, 1).reshape(512, -1)) for i in range(1, len(self.convs))]
        self.zero_bias = torch.zeros(512, dtype=torch.float32, device=dev)
        e = "encoder."
        self.fp_ln = (f32(e + "feature_projection.layer_norm.weight"), f32(e + "feature_projection.layer_norm.bias"))
        self.fp_w, self.fp_b = half(f32(e + "feature_projection.projection.weight")), f32(e + "feature_projection.projection.bias")
        t = e + "transformer."
        g = state.get(prefix + t + "pos_conv_embed.conv.parametrizations.weight.original0", state.get(prefix + t + "pos_conv_embed.conv.weight_g"))
        v = state.get(prefix + t + "pos_conv_embed.conv.parametrizations.weight.original1", state.get(prefix + t + "pos_conv_embed.conv.weight_v"))
        if g is None or v is None:
            raise K.AvexHipError("AVES: positional conv weight-norm parameters missing from the state dict")
        tg = torch.from_numpy(np.ascontiguousarray(np.asarray(g, np.float32))).to(dev)
        tv = torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32))).to(dev)
        self.pc_w = K.posconv_pack(tg, tv, self.G, operand_dtype)
        self.pc_b = f32(t + "pos_conv_embed.conv.bias")
        self.enc_ln = (f32(t + "layer_norm.weight"), f32(t + "layer_norm.bias"))
        self.layers = []
        for i in range(self.L):
            p = t + f"layers.{i}."
            wq, wk, wv = (f32(p + f"attention.{n}.weight") for n in ("q_proj", "k_proj", "v_proj"))
            bq, bk, bv = (f32(p + f"attention.{n}.bias") for n in ("q_proj", "k_proj", "v_proj"))
            self.layers.append(dict(
                w_qkv=half(torch.cat([wq, wk, wv], 0)), b_qkv=torch.cat([bq, bk, bv], 0).contiguous(),
                w_o=half(f32(p + "attention.out_proj.weight")), b_o=f32(p + "attention.out_proj.bias"),
                ln1=(f32(p + "layer_norm.weight"), f32(p + "layer_norm.bias")),
                w1=half(f32(p + "feed_forward.intermediate_dense.weight")), b1=f32(p + "feed_forward.intermediate_dense.bias"),
                w2=half(f32(p + "feed_forward.output_dense.weight")), b2=f32(p + "feed_forward.output_dense.bias"),
                ln2=(f32(p + "final_layer_norm.weight"), f32(p + "final_layer_norm.bias"))))

    def num_tokens(self, T: int) -> int:
        return conv_frame_plan(T, self.convs)[0][-1]

    def extract_conv_features(self, wav: torch.Tensor) -> torch.Tensor:
        """``[B, T]`` -> half ``[B, frames, 512]`` (output of the 7-layer feature extractor)."""
        B, T = wav.shape
        F, P = conv_frame_plan(T, self.convs)
        x = K.wavconv0(wav, self.w0, self.gn_w, self.gn_b, P[0], slack_rows=8, dtype=self.dtype)
        for l in range(1, len(self.convs)):
            _c, k, s = self.convs[l]
            rows = B * P[l]
            x = K.gemm(x, self.wc[l - 1], bias=self.zero_bias, gelu=True, out_f32=False, out_half=True, lda=s * 512, rows=rows, kdim=k * 512,
                       slack_rows=8)["half"]      # slack: the next layer's rows of the last clip read a little past it
        return x[:B * P[-1]].view(B, P[-1], 512)[:, :F[-1]].contiguous()

    @torch.no_grad()
    def forward(self, wav: torch.Tensor, hook_layers: Iterable[int] = (), want_features: bool = True, want_pooled: bool = False
                ) -> Dict[str, object]:
        """``hook_layers``: transformer layer indices whose ``feed_forward.output_dense`` output is returned (fp32 ``[B, T', 768]``)."""
        if wav.dim() != 2 or wav.dtype != torch.float32 or not wav.is_cuda:
            raise ValueError("wav must be a [B, T] float32 CUDA tensor")
        B = wav.shape[0]
        hooks = set(int(i) for i in hook_layers)
        feats = self.extract_conv_features(wav)
        Tt = feats.shape[1]
        M, E = B * Tt, self.E
        _, h = K.layernorm(feats.view(M, 512), *self.fp_ln, want_f32=False)
        x = K.gemm(h, self.fp_w, bias=self.fp_b, out_f32=False, out_half=True)["half"]
        pre = K.posconv(x.view(B, Tt, E), None, self.pc_w, self.pc_b, self.G, self.KP, half_out=True).view(M, E)
        _, x = K.layernorm(pre, *self.enc_ln, want_f32=False)
        out: Dict[str, object] = {"hooks": {}}
        x32 = None
        for i, ly in enumerate(self.layers):
            qkv = K.gemm(x, ly["w_qkv"], bias=ly["b_qkv"], out_f32=False, out_half=True)["half"]
            a = K.attention(qkv, B, Tt, self.H, None, None, None, None)
            pre = K.gemm(a, ly["w_o"], bias=ly["b_o"], resid_half=x, alpha=1.0, out_f32=False, out_half=True)["half"]
            _, x = K.layernorm(pre, *ly["ln1"], want_f32=False)
            hdn = K.gemm(x, ly["w1"], bias=ly["b1"], gelu=True, out_f32=False, out_half=True)["half"]
            r = K.gemm(hdn, ly["w2"], bias=ly["b2"], resid_half=x, alpha=1.0, out_f32=False, out_half=True, out_raw=i in hooks)
            if i in hooks:
                out["hooks"][i] = r["raw"].view(B, Tt, E)
            last = i == self.L - 1
            x32, x = K.layernorm(r["half"], *ly["ln2"], want_f32=last and (want_features or want_pooled), want_half=not last)
        if want_features:
            out["features"] = x32.view(B, Tt, E)
        if want_pooled:
            out["pooled"] = K.mean_pool(x32.view(B, Tt, E))
        return out
