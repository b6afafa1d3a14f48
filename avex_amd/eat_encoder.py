"""EAT-base (Data2Vec-multi image encoder on the log-mel image) on the MI355X kernels.

Backbone arithmetic of the reference's ``EATHFModel`` (avex/models/eat_hf.py:201,241-289: ``backbone.extract_features(spec[B,1,1024,128])``
-> ``(B, 513, 768)``), whose encoder is HF Hub remote code (``worstchan/EAT-base_epoch30_pretrain``) that is absent from the
reference tree: PARITY UNPINNED, the checker is ``oracle/eat_oracle.py`` (its header lists what the reference tree does fix:
input image, output shape, parameter names, hook names).  Everything runs in libavexhip.so:

  waveform --fbank_kernel--> 16 x 16 patch rows (operand type) straight from the filterbank (per-clip mean removed, Hann window,
           rows past the last frame = normalised zero padding: EATAudioProcessor, avex/models/eat/audio_processor.py:72-143)
  --gemm(local_encoder 256 -> 768, + bias)--> --token_embed_ln (class token row, + fixed positions, pre_norm)--> x [B * 513, 768]
  12 x { gemm(qkv 768 -> 2304) ; attention (no bias table: 513 tokens = two query blocks, three key blocks) ;
         gemm(proj) + bias (raw tap = hook ``blocks.{i}.attn.proj``), + x ; LayerNorm(norm1) ;
         gemm(fc1) + GELU ; gemm(fc2) + bias + x ; LayerNorm(norm2) }
  -> features [B, 513, 768] fp32 (the last norm2), CLS / mean pooling.

Activations between kernels are in the operand type (f16 default), accumulation / LayerNorm / softmax statistics fp32, like the
BEATs path.
"""
from __future__ import annotations

from typing import Dict, Iterable, Mapping, Optional

import numpy as np
import torch

from . import _capi
from . import kernels as K

__all__ = ["EatEncoder"]


class EatEncoder:
    """``[B, T]`` fp32 waveforms (or a ``[B, 1024, 128]`` log-mel image) on the GPU -> ``[B, 513, 768]`` features / hook taps."""

    def __init__(self, cfg: Mapping[str, object], state: Mapping[str, object], operand_dtype: str = "f16", prefix: str = "backbone.model.",
                 norm_mean: float = -4.268, norm_std: float = 4.569) -> None:
        _capi.require_gpu()
        self.cfg = dict(cfg)
        self.dtype = operand_dtype
        self.E = int(cfg["embed_dim"]); self.H = int(cfg["num_heads"]); self.L = int(cfg["depth"]); self.P = int(cfg["patch_size"])
        self.target_length = int(cfg["target_length"]); self.n_mels = int(cfg["n_mels"])
        self.eps = float(cfg.get("norm_eps", 1e-6))
        if self.E != 64 * self.H:
            raise K.AvexHipError("EAT: head_dim must be 64")
        if bool(cfg.get("layer_norm_first", False)):
            raise K.AvexHipError("EAT: layer_norm_first=True (pre-LN blocks) is not built")
        if self.P != 16 or self.n_mels % 16 or self.target_length % 16:
            raise K.AvexHipError("EAT: only 16 x 16 patches over a (16 a) x (16 b) image are built")
        dev = torch.device("cuda", torch.cuda.current_device())

        def f32(name: str) -> torch.Tensor:
            v = state[prefix + name]
            if isinstance(v, torch.Tensor):
                return v.detach().to(device=dev, dtype=torch.float32).contiguous()
            return torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32))).to(dev)

        half = lambda t: K.to_half(t.contiguous(), operand_dtype)
        self.n_patches = (self.target_length // 16) * (self.n_mels // 16)
        self.w_pe = half(f32("local_encoder.proj.weight").reshape(self.E, 256)); self.b_pe = f32("local_encoder.proj.bias")
        self.cls = f32("extra_tokens").reshape(self.E)
        pos = f32("fixed_positional_encoder.positions").reshape(-1, self.E)
        if pos.shape[0] < self.n_patches:
            raise K.AvexHipError(f"EAT: position table has {pos.shape[0]} rows, {self.n_patches} patches need one each")
        self.pos = pos[: self.n_patches].contiguous()
        self.pre_norm = (f32("pre_norm.weight"), f32("pre_norm.bias"))
        self.layers = []
        for i in range(self.L):
            p = f"blocks.{i}."
            self.layers.append(dict(
                w_qkv=half(f32(p + "attn.qkv.weight")), b_qkv=f32(p + "attn.qkv.bias"),
                w_o=half(f32(p + "attn.proj.weight")), b_o=f32(p + "attn.proj.bias"),
                ln1=(f32(p + "norm1.weight"), f32(p + "norm1.bias")),
                w1=half(f32(p + "mlp.fc1.weight")), b1=f32(p + "mlp.fc1.bias"),
                w2=half(f32(p + "mlp.fc2.weight")), b2=f32(p + "mlp.fc2.bias"),
                ln2=(f32(p + "norm2.weight"), f32(p + "norm2.bias"))))
        win = 400
        per_sample = norm_mean == 0.0 and norm_std == 1.0
        if per_sample:
            raise K.AvexHipError("EAT: per-sample normalisation (norm_mean 0, norm_std 1) is only built in EATAudioProcessor, not in the fused path")
        self.plan = K.FbankPlan(win_length=win, hop_length=160, n_mels=self.n_mels, input_scale=1.0, preemph=0.97, remove_dc=True,
                                log_floor=K.F32_EPS, norm_mean=float(norm_mean), norm_div=2.0 * float(norm_std), window=K.hann_window(win),
                                mel_fb=K.kaldi_mel_filterbank(self.n_mels, 512, 16000.0, 20.0, 0.0))

    @property
    def num_tokens(self) -> int:
        return self.n_patches + 1

    def patches_from_wav(self, wav: torch.Tensor) -> torch.Tensor:
        return self.plan.patches(wav, out_frames=self.target_length, patch=16, remove_clip_mean=True, dtype=self.dtype)

    def patches_from_spec(self, spec: torch.Tensor) -> torch.Tensor:
        """``[B, target_length, n_mels]`` fp32 image (an EATAudioProcessor output) -> patch rows in the operand type."""
        B = spec.shape[0]
        nt, nf = self.target_length // 16, self.n_mels // 16
        p = spec.reshape(B, nt, 16, nf, 16).permute(0, 1, 3, 2, 4).reshape(B * nt * nf, 256)
        return K.to_half(p.contiguous(), self.dtype)

    @torch.no_grad()
    def forward(self, wav: Optional[torch.Tensor] = None, *, spec: Optional[torch.Tensor] = None, hook_layers: Iterable[int] = (),
                want_features: bool = True, pooling: Optional[str] = None) -> Dict[str, object]:
        """``hook_layers``: block indices whose ``attn.proj`` output is returned (fp32 ``[B, 513, 768]``); ``pooling``: ``"cls"`` /
        ``"mean"`` adds ``out["pooled"]`` ``[B, 768]`` (eat_hf.py:283-288)."""
        if (wav is None) == (spec is None):
            raise ValueError("give exactly one of wav / spec")
        if wav is not None:
            if wav.dim() != 2 or wav.dtype != torch.float32 or not wav.is_cuda:
                raise ValueError("wav must be a [B, T] float32 CUDA tensor")
            B = wav.shape[0]
            patches = self.patches_from_wav(wav.contiguous())
        else:
            if spec.dim() != 3 or tuple(spec.shape[1:]) != (self.target_length, self.n_mels) or not spec.is_cuda:
                raise ValueError(f"spec must be a [B, {self.target_length}, {self.n_mels}] CUDA tensor")
            B = spec.shape[0]
            patches = self.patches_from_spec(spec.float())
        hooks = set(int(i) for i in hook_layers)
        E, Tt = self.E, self.num_tokens
        pe = K.gemm(patches, self.w_pe, bias=self.b_pe, out_f32=False, out_half=True)["half"]
        x, _ = K.token_embed_ln(pe, self.pos, self.cls, *self.pre_norm, self.eps, B)
        out: Dict[str, object] = {"hooks": {}}
        x32 = None
        for i, ly in enumerate(self.layers):
            qkv = K.gemm(x, ly["w_qkv"], bias=ly["b_qkv"], out_f32=False, out_half=True)["half"]
            a = K.attention(qkv, B, Tt, self.H, None, None, None, None)
            r = K.gemm(a, ly["w_o"], bias=ly["b_o"], resid_half=x, alpha=1.0, out_f32=False, out_half=True, out_raw=i in hooks)
            if i in hooks:
                out["hooks"][i] = r["raw"].view(B, Tt, E)
            _, x = K.layernorm(r["half"], *ly["ln1"], eps=self.eps, want_f32=False)
            hdn = K.gemm(x, ly["w1"], bias=ly["b1"], gelu=True, out_f32=False, out_half=True)["half"]
            y = K.gemm(hdn, ly["w2"], bias=ly["b2"], resid_half=x, alpha=1.0, out_f32=False, out_half=True)["half"]
            last = i == self.L - 1
            x32, x = K.layernorm(y, *ly["ln2"], eps=self.eps, want_f32=last, want_half=not last)
        feats = x32.view(B, Tt, E)
        if want_features:
            out["features"] = feats
        if pooling == "cls":
            out["pooled"] = feats[:, 0].contiguous()
        elif pooling == "mean":
            out["pooled"] = K.mean_pool(feats)
        elif pooling is not None:
            raise ValueError("pooling must be 'cls' or 'mean'")
        return out
