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

import ctypes as C
from typing import Dict, Iterable, Mapping, Optional

import torch

from . import _capi
from . import kernels as K

__all__ = ["EatEncoder"]


class EatEncoder:
    """``[B, T]`` fp32 waveforms (or a ``[B, 1024, 128]`` log-mel image) on the GPU -> ``[B, 513, 768]`` features / hook taps.
    A thin wrapper over the ``avexhip_eat`` handle (csrc/encoders.cpp): the library owns the weights, this class the output tensors
    and the workspace."""

    def __init__(self, cfg: Mapping[str, object], state: Mapping[str, object], operand_dtype: str = "f16", prefix: str = "backbone.model.",
                 norm_mean: float = -4.268, norm_std: float = 4.569, max_chunk_clips: int = 0, residual: str = "auto", batch_invariant: bool = False) -> None:
        """``residual``: the inter-kernel residual stream.  ``"half"`` = operand type (LayerNorms folded into the GEMMs), ``"f32"`` = fp32
        stream (un-averaged rows 3 x closer to the fp32 arithmetic, ~20 % more time), ``"auto"`` (default) = per call, BEATs' policy
        (beats_model.py here): ``"f32"`` whenever the call hands back un-averaged rows -- features, un-pooled taps, the class token
        (``pooling="cls"``, the reference wrapper's default: eat_hf.py:149,281-282) -- and ``"half"`` for token-mean outputs only
        (config C3's path).  Two library handles then, each built on first use."""
        _capi.require_gpu()
        self.residual = str(residual or "auto").lower()
        if self.residual != "auto":
            K.residual_code(self.residual)      # validates
        self.cfg = dict(cfg)
        self.dtype = operand_dtype
        self.E = int(cfg["embed_dim"]); self.H = int(cfg["num_heads"]); self.L = int(cfg["depth"]); self.P = int(cfg["patch_size"])
        self.target_length = int(cfg["target_length"]); self.n_mels = int(cfg["n_mels"])
        self.eps = float(cfg.get("norm_eps", 1e-6))
        if bool(cfg.get("layer_norm_first", False)):
            raise K.AvexHipError("EAT: layer_norm_first=True (pre-LN blocks) is not built")
        self.n_patches = (self.target_length // 16) * (self.n_mels // 16) if self.P == 16 else 0
        c = _capi.EatConfig()
        c.embed_dim, c.num_heads, c.depth = self.E, self.H, self.L
        c.ffn_dim = int(cfg.get("ffn_dim", int(float(cfg.get("mlp_ratio", 4.0)) * self.E)))
        c.patch_size, c.target_length, c.n_mels = self.P, self.target_length, self.n_mels
        c.norm_eps, c.norm_mean, c.norm_std = self.eps, float(norm_mean), float(norm_std)
        c.operand_dtype = _capi.dtype_code(operand_dtype)
        c.max_chunk_clips = int(max_chunk_clips)
        self._c, self._batch_invariant = c, bool(batch_invariant)
        self._sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)} if prefix else dict(state)
        self._handles: Dict[str, int] = {}      # residual mode ("half" / "f32") -> library handle
        self._profiling = False
        self._h = self._handle_for(frames=self.residual in ("auto", "f32", "fp32", "float32"))      # a bad checkpoint fails here, not in the first forward
        self._ws: Optional[torch.Tensor] = None
        self._norm = (float(norm_mean), float(norm_std))
        self._plan = None

    def _handle_for(self, frames: bool) -> int:
        """The handle whose residual stream this call wants (``frames``: it returns un-averaged rows); ``self._h`` = the last one used."""
        mode = ("f32" if frames else "half") if self.residual == "auto" else ("half" if K.residual_code(self.residual) & 1 else "f32")
        h = self._handles.get(mode)
        if h is None:
            self._c.residual_dtype = K.residual_code(mode, self._batch_invariant)
            arr, n, keep = K.tensor_table(self._sub)
            h = _capi.lib().avexhip_eat_create(C.byref(self._c), arr, n)
            del keep
            if not h:
                raise K.AvexHipError(f"eat_create failed: {_capi.last_error()}")
            self._handles[mode] = h
            if self._profiling:
                _capi.check(_capi.lib().avexhip_eat_set_profiling(h, 1), "eat_set_profiling")
        self._h = h
        return h

    @property
    def num_tokens(self) -> int:
        return int(_capi.lib().avexhip_eat_num_tokens(self._h))

    @property
    def plan(self) -> "K.FbankPlan":
        """The frontend as a stand-alone plan (the handle has its own copy): for callers that want the patch rows themselves."""
        if self._plan is None:
            self._plan = K.FbankPlan(win_length=400, hop_length=160, n_mels=self.n_mels, input_scale=1.0, preemph=0.97, remove_dc=True,
                                     log_floor=K.F32_EPS, norm_mean=self._norm[0], norm_div=2.0 * self._norm[1], window=K.hann_window(400),
                                     mel_fb=K.kaldi_mel_filterbank(self.n_mels, 512, 16000.0, 20.0, 0.0))
        return self._plan

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
                want_features: bool = True, pooling: Optional[str] = None, hook_pooled=False) -> Dict[str, object]:
        """``hook_layers``: block indices whose ``attn.proj`` output is returned (fp32 ``[B, 513, 768]``, or ``[B, 768]`` token means with
        ``hook_pooled``); ``pooling``: ``"cls"`` / ``"mean"`` adds ``out["pooled"]`` ``[B, 768]`` (eat_hf.py:283-288)."""
        if (wav is None) == (spec is None):
            raise ValueError("give exactly one of wav / spec")
        if pooling not in (None, "cls", "mean"):
            raise ValueError("pooling must be 'cls' or 'mean'")
        if wav is not None:
            if wav.dim() != 2 or wav.dtype != torch.float32 or not wav.is_cuda:
                raise ValueError("wav must be a [B, T] float32 CUDA tensor")
            if wav.stride(1) != 1:
                wav = wav.contiguous()
            B, T, dev = wav.shape[0], wav.shape[1], wav.device
        else:
            if spec.dim() != 3 or tuple(spec.shape[1:]) != (self.target_length, self.n_mels) or not spec.is_cuda:
                raise ValueError(f"spec must be a [B, {self.target_length}, {self.n_mels}] CUDA tensor")
            spec = spec.float().contiguous()
            B, T, dev = spec.shape[0], 0, spec.device
        hook_layers = list(hook_layers)
        frames = bool(want_features or pooling == "cls" or (hook_layers and K.pool_code(hook_pooled) != 1))      # anything but token means
        self._handle_for(frames)
        E, Tt = self.E, self.num_tokens
        need = int(_capi.lib().avexhip_eat_workspace_bytes(self._h, B))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        hooks: Dict[int, torch.Tensor] = {}
        ptrs = (C.c_void_p * max(self.L, 1))()
        mask = 0
        for i in sorted(set(int(x) for x in hook_layers)):
            if not 0 <= i < self.L:
                raise ValueError(f"hook layer {i} out of range 0..{self.L - 1}")
            hooks[i] = torch.empty((B, E) if K.pool_code(hook_pooled) else (B, Tt, E), dtype=torch.float32, device=dev)      # sized by the code the library gets
            ptrs[i] = int(hooks[i].data_ptr())
            mask |= 1 << i
        feats = torch.empty((B, Tt, E), dtype=torch.float32, device=dev) if want_features else None
        pooled = torch.empty((B, E), dtype=torch.float32, device=dev) if pooling else None
        _capi.check(_capi.lib().avexhip_eat_forward(self._h, K._ptr(wav), B, T, wav.stride(0) if wav is not None else 0, K._ptr(spec), mask, ptrs,
                                                    K.pool_code(hook_pooled), K._ptr(feats), K._ptr(pooled), {None: 0, "cls": 1, "mean": 2}[pooling],
                                                    K._ptr(self._ws), self._ws.numel(), K._stream()), "eat_forward")
        out: Dict[str, object] = {"hooks": hooks}
        if want_features:
            out["features"] = feats
        if pooling:
            out["pooled"] = pooled
        return out

    def overflow_events(self, sync: bool = True) -> int:
        total = 0
        for h in self._handles.values():
            n = C.c_uint32(0)
            _capi.check(_capi.lib().avexhip_eat_overflow_count(h, C.byref(n), K._stream(), int(bool(sync))), "eat_overflow_count")
            total += int(n.value)
        return total

    def set_profiling(self, enabled: bool) -> None:
        self._profiling = bool(enabled)
        for h in self._handles.values():
            _capi.check(_capi.lib().avexhip_eat_set_profiling(h, int(enabled)), "eat_set_profiling")

    def last_profile(self):
        return K.handle_profile(_capi.lib().avexhip_eat_last_profile, self._h)

    def close(self) -> None:
        for h in getattr(self, "_handles", {}).values():
            _capi.lib().avexhip_eat_destroy(h)
        self._handles = {}
        self._h = None
        self._ws = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
