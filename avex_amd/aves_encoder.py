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

import ctypes as C
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
    """``[B, T]`` fp32 waveforms on the GPU -> last-layer features ``[B, T', 768]`` / hook taps / pooled embeddings.
    A thin wrapper over the ``avexhip_aves`` handle (csrc/encoders.cpp): the library owns the weights, this class the output tensors
    and the workspace."""

    def __init__(self, cfg: Mapping[str, object], state: Mapping[str, np.ndarray], operand_dtype: str = "f16", prefix: str = "model.",
                 max_chunk_clips: int = 0, residual: str = "auto", batch_invariant: bool = False) -> None:
        """``residual``: ``"half"`` (operand-type residual stream), ``"f32"`` or ``"auto"`` (default; BEATs' policy, beats_model.py here):
        the fp32 stream for calls that hand back un-averaged rows (features, un-pooled taps: what the reference wrapper's ``forward`` and
        hooks return, aves_model.py:129-150), the operand-type stream for token-mean outputs only.  Two library handles then, each built
        on first use."""
        _capi.require_gpu()
        self.residual = str(residual or "auto").lower()
        if self.residual != "auto":
            K.residual_code(self.residual)      # validates
        self.cfg = dict(cfg)
        self.dtype = operand_dtype
        self.convs = [tuple(int(v) for v in c) for c in cfg["extractor_conv_layer_config"]]
        if self.convs[0] != (512, 10, 5) or any(c[0] != 512 for c in self.convs) or len(self.convs) > 8:
            raise K.AvexHipError("AVES feature extractor: only the wav2vec2-base layout (512 channels, first layer k=10 s=5) is built")
        if bool(cfg.get("encoder_layer_norm_first", False)):
            raise K.AvexHipError("AVES: encoder_layer_norm_first=True (pre-LN) is not built")
        self.E = int(cfg["encoder_embed_dim"]); self.H = int(cfg["encoder_num_heads"]); self.L = int(cfg["encoder_num_layers"])
        self.G = int(cfg["encoder_pos_conv_groups"]); self.KP = int(cfg["encoder_pos_conv_kernel"])
        c = _capi.AvesConfig()
        c.embed_dim, c.num_heads, c.num_layers = self.E, self.H, self.L
        c.ffn_dim = int(cfg.get("encoder_ff_interm_features", 4 * self.E))
        c.pos_conv_kernel, c.pos_conv_groups, c.n_conv_layers = self.KP, self.G, len(self.convs)
        for i, (_c, k, st) in enumerate(self.convs):
            c.conv_kernel[i], c.conv_stride[i] = k, st
        c.operand_dtype = _capi.dtype_code(operand_dtype)
        c.max_chunk_clips = int(max_chunk_clips)
        self._c, self._batch_invariant = c, bool(batch_invariant)
        self._sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)} if prefix else dict(state)
        self._handles: Dict[str, int] = {}      # residual mode ("half" / "f32") -> library handle
        self._profiling = False
        self._h = self._handle_for(frames=self.residual in ("auto", "f32", "fp32", "float32"))      # a bad checkpoint fails here, not in the first forward
        self._ws: Optional[torch.Tensor] = None
        self._state, self._prefix = state, prefix
        self._conv_weights = None

    def _handle_for(self, frames: bool) -> int:
        """The handle whose residual stream this call wants (``frames``: it returns un-averaged rows); ``self._h`` = the last one used."""
        mode = ("f32" if frames else "half") if self.residual == "auto" else ("half" if K.residual_code(self.residual) & 1 else "f32")
        h = self._handles.get(mode)
        if h is None:
            self._c.residual_dtype = K.residual_code(mode, self._batch_invariant)
            arr, n, keep = K.tensor_table(self._sub)
            h = _capi.lib().avexhip_aves_create(C.byref(self._c), arr, n)
            del keep
            if not h:
                raise K.AvexHipError(f"aves_create failed: {_capi.last_error()}")
            self._handles[mode] = h
            if self._profiling:
                _capi.check(_capi.lib().avexhip_aves_set_profiling(h, 1), "aves_set_profiling")
        self._h = h
        return h

    def num_tokens(self, T: int) -> int:
        return conv_frame_plan(T, self.convs)[0][-1]

    def extract_conv_features(self, wav: torch.Tensor) -> torch.Tensor:
        """``[B, T]`` -> half ``[B, frames, 512]``: the 7-layer feature extractor alone, composed from the exported kernels (layer 0 =
        ``avexhip_wavconv0``, layers 1-6 = strided-row ``avexhip_gemm``) -- what the handle runs as its first stage, kept for tests."""
        if self._conv_weights is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            def f32(name):
                v = self._state[self._prefix + name]
                if isinstance(v, torch.Tensor):
                    return v.detach().to(device=dev, dtype=torch.float32).contiguous()
                return torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32))).to(dev)
            fe = "feature_extractor.conv_layers."
            self._conv_weights = dict(
                w0=f32(fe + "0.conv.weight").reshape(512, 10), gn_w=f32(fe + "0.layer_norm.weight"), gn_b=f32(fe + "0.layer_norm.bias"),
                # conv weights [out, in, k] -> [out, k, in]: the K order of a strided activation row is (frame, channel)
                wc=[K.to_half(f32(fe + f"{i}.conv.weight").permute(0, 2, 1).reshape(512, -1).contiguous(), self.dtype) for i in range(1, len(self.convs))],
                zero_bias=torch.zeros(512, dtype=torch.float32, device=dev))
        cw = self._conv_weights
        B, T = wav.shape
        F, P = conv_frame_plan(T, self.convs)
        x = K.wavconv0(wav, cw["w0"], cw["gn_w"], cw["gn_b"], P[0], slack_rows=8, dtype=self.dtype)
        for l in range(1, len(self.convs)):
            _c, k, s = self.convs[l]
            rows = B * P[l]
            x = K.gemm(x, cw["wc"][l - 1], bias=cw["zero_bias"], gelu=True, out_f32=False, out_half=True, lda=s * 512, rows=rows, kdim=k * 512,
                       slack_rows=8)["half"]      # slack: the next layer's rows of the last clip read a little past it
        return x[:B * P[-1]].view(B, P[-1], 512)[:, :F[-1]].contiguous()

    @torch.no_grad()
    def forward(self, wav: torch.Tensor, hook_layers: Iterable[int] = (), want_features: bool = True, want_pooled: bool = False,
                frame_pad: Optional[torch.Tensor] = None, hook_pooled=False) -> Dict[str, object]:
        """``hook_layers``: transformer layer indices whose ``feed_forward.output_dense`` output is returned (fp32 ``[B, T', 768]``)."""
        if wav.dim() != 2 or wav.dtype != torch.float32 or not wav.is_cuda:
            raise ValueError("wav must be a [B, T] float32 CUDA tensor")
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, T = wav.shape
        dev = wav.device
        Tt = self.num_tokens(T)
        E = self.E
        hook_layers = list(hook_layers)
        self._handle_for(frames=bool(want_features or (hook_layers and K.pool_code(hook_pooled) != 1)))      # anything but token means
        need = int(_capi.lib().avexhip_aves_workspace_bytes(self._h, B, T))
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
        pooled = torch.empty((B, E), dtype=torch.float32, device=dev) if want_pooled else None
        pad = None
        if frame_pad is not None:
            pad = frame_pad.to(device=dev, dtype=torch.uint8).contiguous()
            if pad.shape != (B, Tt):
                raise ValueError(f"frame_pad must be [B={B}, T'={Tt}], got {tuple(pad.shape)}")
        _capi.check(_capi.lib().avexhip_aves_forward(self._h, K._ptr(wav), B, T, wav.stride(0), K._ptr(pad), mask, ptrs, K.pool_code(hook_pooled), K._ptr(feats),
                                                     K._ptr(pooled), K._ptr(self._ws), self._ws.numel(), K._stream()), "aves_forward")
        out: Dict[str, object] = {"hooks": hooks}
        if want_features:
            out["features"] = feats
        if want_pooled:
            out["pooled"] = pooled
        return out

    def overflow_events(self, sync: bool = True) -> int:
        total = 0
        for h in self._handles.values():
            n = C.c_uint32(0)
            _capi.check(_capi.lib().avexhip_aves_overflow_count(h, C.byref(n), K._stream(), int(bool(sync))), "aves_overflow_count")
            total += int(n.value)
        return total

    def set_profiling(self, enabled: bool) -> None:
        self._profiling = bool(enabled)
        for h in self._handles.values():
            _capi.check(_capi.lib().avexhip_aves_set_profiling(h, int(enabled)), "aves_set_profiling")

    def last_profile(self):
        return K.handle_profile(_capi.lib().avexhip_aves_last_profile, self._h)

    def close(self) -> None:
        for h in getattr(self, "_handles", {}).values():
            _capi.lib().avexhip_aves_destroy(h)
        self._handles = {}
        self._h = None
        self._ws = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
