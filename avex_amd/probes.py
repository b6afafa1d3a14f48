"""Probe heads evaluated on the device: forward-only mirrors of the reference's online probes.

Reference: ``avex/models/probes/{base_probes,linear_probe,mlp_probe,attention_probe,transformer_probe,lstm_probe}.py``.  The reference probes are
``nn.Module`` s trained by autograd on top of ``base_model.extract_embeddings``; with this package's models that call already returns
device tensors, so the reference's own probe classes train on them unchanged and nothing crosses PCIe.  What is mirrored here is the
EVALUATION forward -- layer mixing (``_sum``), the 2-D / 3-D reshaping rules and the three heads -- running on this library's fp32
kernels (``avexhip_layer_mix``, ``avexhip_dense_f32``, ``avexhip_mha_f32``, ``avexhip_layernorm``, ``avexhip_mean_pool``) with the
reference's constructor arguments and ``state_dict`` key names, so a probe trained with the reference loads with
``load_state_dict`` and scores batches right behind the encoder.  Training (``.train()``) is refused loudly; embedding projectors
(taps of different widths, base_probes.py:262-289) are not built: every model on this path has equal-width taps.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import kernels as K

TensorOrList = Union[torch.Tensor, List[torch.Tensor]]


class _DeviceProbe(nn.Module):
    """Shared scaffolding (base_probes.py:20-206): where the embeddings come from and how several taps are combined."""

    rank = 2                       # 2: (batch, features) heads; 3: (batch, sequence, features) heads

    def __init__(self, base_model, layers: Sequence[str], num_classes: int, device: str = "cuda", feature_mode: bool = False,
                 input_dim=None, aggregation: str = "mean", target_length: Optional[int] = None, freeze_backbone: bool = True) -> None:
        super().__init__()
        if not freeze_backbone:
            raise NotImplementedError("avex_amd probes are evaluation heads on a frozen backbone")
        if not feature_mode and base_model is None:
            raise ValueError("a base_model is required unless feature_mode=True")
        if feature_mode and base_model is None and input_dim is None:
            raise ValueError("input_dim must be provided when feature_mode=True and base_model is None")
        self.device = device
        self.base_model = base_model
        self.layers = list(layers)
        self.num_classes = num_classes
        self.feature_mode = feature_mode
        self.aggregation = aggregation
        self.target_length = target_length
        self.freeze_backbone = True
        shapes = self._probe_input_shapes(input_dim)
        self.embedding_projectors: Optional[nn.ModuleList] = None
        if len(shapes) > 1 or self._list_input:                     # a LIST of embeddings: projectors + layer weights
            self.inferred_dim = self._analyze_and_create_projectors(shapes)
            if len(shapes) > 1:                                     # base_probes.py:150-152: only for several embeddings
                self.register_buffer("layer_weights", torch.zeros(len(shapes)))
        else:
            self.inferred_dim = self._feature_dim(shapes[0])
        self.build_head(self.inferred_dim)
        self.to(device)
        super().train(False)

    # -- construction ------------------------------------------------------------------------------------------------------
    def _analyze_and_create_projectors(self, shapes: List[tuple]) -> int:
        """Which taps need an ``nn.Linear`` to the common width (base_probes.py:254-289 for 2-D probes, :333-367 for 3-D ones): the
        target is the width (and, for 3-D, the sequence length) more than half of the taps share, else the largest; taps that differ
        get a projector (3-D: also when only their sequence length differs, as the reference does).  Entries are ``None`` where no
        projection is needed, so ``state_dict`` keys are ``embedding_projectors.{i}.weight/bias`` like the reference's."""
        from collections import Counter
        if self.rank == 2:
            dims = [self._feature_dim(s) for s in shapes]
            common, count = Counter(dims).most_common(1)[0]
            target = common if count > len(dims) / 2 else max(dims)
            need = [d != target for d in dims]
            in_dims = dims
        else:
            info = []
            for s in shapes:                                        # (seq_len, feat) per tap, shapes are without the batch dim
                if len(s) == 2:
                    info.append((s[0], s[1]))
                elif len(s) == 3:
                    info.append((s[2], s[0] * s[1]))
                elif len(s) == 1:
                    info.append((s[0], 1))
                else:
                    raise ValueError(f"Unsupported embedding dim {len(s) + 1} for 3D probe")
            seq_c, seq_n = Counter(q for q, _ in info).most_common(1)[0]
            feat_c, feat_n = Counter(f for _, f in info).most_common(1)[0]
            target_seq = seq_c if seq_n > len(info) / 2 else max(q for q, _ in info)
            target = feat_c if feat_n > len(info) / 2 else max(f for _, f in info)
            need = [not (f == target and q == target_seq) for q, f in info]
            in_dims = [f for _, f in info]
        self.embedding_projectors = nn.ModuleList([nn.Linear(d, target) if n else None for d, n in zip(in_dims, need)])
        for m in self.embedding_projectors:
            if m is not None:
                for p_ in m.parameters():
                    p_.requires_grad_(False)
        return target

    def _probe_input_shapes(self, input_dim) -> List[tuple]:
        """Per-embedding shapes without the batch dim (base_probes.py:93-163)."""
        self._list_input = False
        if self.feature_mode and input_dim is not None:
            if isinstance(input_dim, list):
                self._list_input = len(input_dim) > 1
                return [tuple(s) for s in input_dim]
            if isinstance(input_dim, tuple):
                return [input_dim]
            return [(1, int(input_dim))] if self.rank == 3 else [(int(input_dim),)]
        ap = self.base_model.audio_processor
        if self.target_length is not None:
            n = int(self.target_length)
        elif hasattr(ap, "target_length_seconds"):
            n = int(ap.target_length_seconds * ap.sr)
        elif hasattr(ap, "target_length"):
            n = int(ap.target_length)
        else:
            raise ValueError("target_length must be provided when base_model.audio_processor does not have target_length or "
                             "target_length_seconds")
        with torch.no_grad():
            emb = self.base_model.extract_embeddings(torch.randn(1, n, device=self.device), aggregation=self.aggregation)
        self._list_input = isinstance(emb, list)
        return [tuple(e.shape[1:]) for e in (emb if isinstance(emb, list) else [emb])]

    def _feature_dim(self, shape: tuple) -> int:
        n = 1
        if self.rank == 2:                                          # base_probes.py:263-272,291-298
            for s in shape:
                n *= s
            return n
        if len(shape) == 2:                                         # base_probes.py:336-343,380-387
            return shape[1]
        if len(shape) == 3:
            return shape[0] * shape[1]
        if len(shape) == 1:
            return 1
        raise ValueError(f"unsupported embedding shape {shape}")

    def build_head(self, inferred_dim: int) -> None:
        raise NotImplementedError

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("avex_amd probes are forward-only; train the reference's probe class on this model's "
                                      "device-resident embeddings and load its state_dict here")
        return super().train(False)

    # -- embeddings --------------------------------------------------------------------------------------------------------
    def _get_embeddings(self, x, padding_mask) -> TensorOrList:
        """base_probes.py:166-195."""
        if self.feature_mode:
            if isinstance(x, dict):
                if "raw_wav" in x:
                    return x["raw_wav"]
                keys = [k for k in x if k not in ("label", "padding_mask")]
                return x[keys[0]] if len(keys) == 1 else [x[k] for k in keys]
            return x
        if isinstance(x, dict):
            padding_mask = x.get("padding_mask")
            x = x["raw_wav"]
        return self.base_model.extract_embeddings(x, padding_mask=padding_mask, aggregation=self.aggregation)

    def _mix(self, taps: List[torch.Tensor]) -> torch.Tensor:
        return K.layer_mix([t.float() for t in taps], getattr(self, "layer_weights", None))

    def _project(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """``embedding_projectors[i](x)`` on the device (fp32 MFMA); ``x`` is ``[B, in]`` or ``[B, T, in]``."""
        proj = self.embedding_projectors[i] if self.embedding_projectors is not None and i < len(self.embedding_projectors) else None
        if proj is None:
            return x
        lead = x.shape[:-1]
        y = K.dense_f32(x.reshape(-1, x.shape[-1]).float().contiguous(), proj.weight.detach(), proj.bias.detach())
        return y.reshape(*lead, y.shape[-1])

    def _combine(self, emb: TensorOrList) -> torch.Tensor:
        if self.rank == 2:                                          # base_probes.py:299-322
            if isinstance(emb, list):
                return self._mix([self._project(i, e.reshape(e.shape[0], -1)) for i, e in enumerate(emb)])
            return emb.reshape(emb.shape[0], -1).float()
        fmt = self._seq_feat                                        # base_probes.py:389-414
        if isinstance(emb, list):
            taps = [self._project(i, fmt(e)) for i, e in enumerate(emb)]
            lens = [t.shape[1] for t in taps]
            if len(set(lens)) > 1:                                  # to the SHORTEST length, linear, align_corners=False (:398-411)
                tgt = min(lens)
                taps = [t if t.shape[1] == tgt else K.seq_interp_linear(t, tgt) for t in taps]
            return self._mix(taps)
        return fmt(emb).float()

    @staticmethod
    def _seq_feat(e: torch.Tensor) -> torch.Tensor:                 # base_probes.py:369-378
        if e.dim() == 3:
            return e
        if e.dim() == 4:
            b, c, h, w = e.shape
            return e.transpose(1, 3).reshape(b, w, c * h)
        if e.dim() == 2:
            return e.unsqueeze(2)
        raise ValueError(f"Unsupported embedding dim {e.dim()} for 3D probe")

    def get_learned_weights_table(self) -> str:                     # base_probes.py:208-244
        lw = getattr(self, "layer_weights", None)
        if lw is None:
            return "No learned weights found. This probe does not use weighted sum of embeddings."
        raw = lw.detach().cpu()
        norm = torch.softmax(raw, dim=0)
        rows = ["Learned Layer Weights:", "=" * 50, f"{'Layer':<15} {'Raw Weight':<12} {'Normalized':<12} {'Percentage':<12}", "-" * 50]
        rows += [f"{'Layer_%d' % i:<15} {r:<12.4f} {n:<12.4f} {n * 100:<12.2f}%" for i, (r, n) in enumerate(zip(raw.tolist(), norm.tolist()))]
        rows += ["-" * 50, "Sum of normalized weights: %.6f" % float(norm.sum()), "Number of layers: %d" % len(raw)]
        return "\n".join(rows)

    def _buf(self, name: str, *shape: int) -> None:
        mod, _, leaf = name.rpartition(".")
        owner = self
        for part in mod.split(".") if mod else []:
            if not hasattr(owner, part):
                owner.add_module(part, nn.Module())
            owner = getattr(owner, part)
        owner.register_buffer(leaf, torch.zeros(*shape))

    def _p(self, name: str) -> torch.Tensor:
        return self.get_buffer(name)

    # -- the encoders' half-precision layer kernels (avexhip_stack_*) for the sequence probes' attention layers --
    @staticmethod
    def _stack_ok(d: int, heads: int, ffn: int) -> bool:
        """Shapes the layer-stack handle takes: head width 32 / 64 / 96 / 128, widths multiples of 128 (``ffn`` 0 = attention-only
        blocks).  ``AVEX_AMD_PROBE_PRECISION=fp32`` keeps the fp32 composition."""
        import os
        return (os.environ.get("AVEX_AMD_PROBE_PRECISION", "half") != "fp32" and d % heads == 0 and d // heads in (32, 64, 96, 128)
                and d % 128 == 0 and ffn % 128 == 0)

    def _stack_table(self) -> dict:                                 # avexhip_stack's parameter name -> this module's buffer name
        raise NotImplementedError

    def _stack_forward(self, h: torch.Tensor, padding_mask: Optional[torch.Tensor], heads: int, layers: int, ffn: int) -> torch.Tensor:
        """``layers`` post-LN blocks on f16 operands with fp32 accumulation (the transformer probe's default width, 768 / 12 heads / 4
        layers: 4.6 ms per 256 x 496 tokens instead of 46 ms on the fp32 kernels; logits within 1e-3 of them)."""
        import ctypes as C
        from . import _capi
        names = self._stack_table()
        version = tuple(self.get_buffer(n)._version for n in names.values()) + (str(h.device),)
        if getattr(self, "_stack", None) is None or self._stack_version != version:
            self._stack_close()
            cfg = _capi.StackConfig()
            cfg.embed_dim, cfg.num_heads, cfg.num_layers, cfg.ffn_dim = h.shape[-1], heads, layers, ffn
            cfg.norm_eps, cfg.activation, cfg.operand_dtype, cfg.max_chunk_clips, cfg.residual_dtype = 1e-5, 3, _capi.F16, 0, 1
            table = {k: self.get_buffer(n).detach().to(device=h.device, dtype=torch.float32).contiguous() for k, n in names.items()}
            arr, n, keep = K.tensor_table(table)
            self._stack = _capi.lib().avexhip_stack_create(C.byref(cfg), arr, n)
            del keep
            if not self._stack:
                raise _capi.AvexHipError(f"stack_create failed: {_capi.last_error()}")
            self._stack_version = version
        B, T, D = h.shape
        need = int(_capi.lib().avexhip_stack_workspace_bytes(self._stack, B, T))
        ws = getattr(self, "_stack_ws", None)
        if ws is None or ws.numel() < need or ws.device != h.device:
            self._stack_ws = ws = torch.empty((need,), dtype=torch.uint8, device=h.device)
        pad = None if padding_mask is None else padding_mask.to(device=h.device, dtype=torch.uint8).contiguous()
        out = torch.empty_like(h)
        _capi.check(_capi.lib().avexhip_stack_forward(self._stack, K._ptr(h), B, T, K._ptr(pad), K._ptr(out), None, K._ptr(ws), ws.numel(), K._stream()),
                    "stack_forward")
        return out

    def _stack_close(self) -> None:
        if getattr(self, "_stack", None):
            from . import _capi
            _capi.lib().avexhip_stack_destroy(self._stack)
        self._stack = None

    def __del__(self) -> None:
        try:
            self._stack_close()
        except Exception:  # noqa: BLE001
            pass


class LinearProbe(_DeviceProbe):
    """linear_probe.py:16-68: ``classifier = Linear(inferred_dim, num_classes)`` on the combined embedding."""

    def build_head(self, inferred_dim: int) -> None:
        self._buf("classifier.weight", self.num_classes, inferred_dim)
        self._buf("classifier.bias", self.num_classes)

    @torch.no_grad()
    def forward(self, x, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        emb = self._combine(self._get_embeddings(x, padding_mask))
        return K.dense_f32(emb, self._p("classifier.weight"), self._p("classifier.bias"))


class MLPProbe(_DeviceProbe):
    """mlp_probe.py:16-91: ``Linear -> activation -> Dropout`` per hidden width, then ``Linear`` to the classes.  State-dict indices
    follow the reference's ``nn.Sequential`` (a Dropout slot exists only for ``dropout_rate > 0``)."""

    def __init__(self, base_model, layers, num_classes, device="cuda", feature_mode=False, input_dim=None, aggregation="mean",
                 hidden_dims: Optional[List[int]] = None, dropout_rate: float = 0.1, activation: str = "relu", target_length=None,
                 freeze_backbone=True) -> None:
        self.hidden_dims = hidden_dims or [512, 256]
        self.dropout_rate = dropout_rate
        self.activation = activation
        if activation not in ("relu", "gelu", "tanh"):
            raise ValueError(f"Unsupported activation: {activation}")
        super().__init__(base_model, layers, num_classes, device, feature_mode, input_dim, aggregation, target_length, freeze_backbone)

    def build_head(self, inferred_dim: int) -> None:
        self._linear_slots: List[int] = []
        slot, cur = 0, inferred_dim
        for h in self.hidden_dims:
            self._buf(f"mlp.{slot}.weight", h, cur)
            self._buf(f"mlp.{slot}.bias", h)
            self._linear_slots.append(slot)
            slot += 3 if self.dropout_rate > 0 else 2
            cur = h
        self._buf(f"mlp.{slot}.weight", self.num_classes, cur)
        self._buf(f"mlp.{slot}.bias", self.num_classes)
        self._linear_slots.append(slot)

    @torch.no_grad()
    def forward(self, x, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self._combine(self._get_embeddings(x, padding_mask))
        for n, slot in enumerate(self._linear_slots):
            last = n + 1 == len(self._linear_slots)
            h = K.dense_f32(h, self._p(f"mlp.{slot}.weight"), self._p(f"mlp.{slot}.bias"), act=None if last else self.activation)
        return h


class AttentionProbe(_DeviceProbe):
    """attention_probe.py:16-134: optional sinusoidal positions, ``num_layers`` x [self attention -> LayerNorm(x + attn)], mean over
    the sequence, classifier.  ``embed_dim`` is the tap width (``attention_dim`` is carried but unused, as in the reference)."""

    rank = 3

    def __init__(self, base_model, layers, num_classes, device="cuda", feature_mode=False, input_dim=None, aggregation="mean",
                 num_heads: int = 8, attention_dim: int = 512, num_layers: int = 2, dropout_rate: float = 0.1,
                 max_sequence_length: Optional[int] = None, use_positional_encoding: bool = False, target_length=None,
                 freeze_backbone=True) -> None:
        self.num_heads = num_heads
        self.attention_dim = attention_dim
        self.num_layers = num_layers
        self.dropout_rate = dropout_rate
        self.max_sequence_length = max_sequence_length
        self.use_positional_encoding = use_positional_encoding
        super().__init__(base_model, layers, num_classes, device, feature_mode, input_dim, aggregation, target_length, freeze_backbone)

    def build_head(self, d: int) -> None:
        if d % self.num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        for i in range(self.num_layers):
            self._buf(f"attention_layers.{i}.in_proj_weight", 3 * d, d)
            self._buf(f"attention_layers.{i}.in_proj_bias", 3 * d)
            self._buf(f"attention_layers.{i}.out_proj.weight", d, d)
            self._buf(f"attention_layers.{i}.out_proj.bias", d)
            self._buf(f"layer_norms.{i}.weight", d)
            self._buf(f"layer_norms.{i}.bias", d)
            self.get_buffer(f"layer_norms.{i}.weight").fill_(1.0)
        if self.use_positional_encoding:                           # attention_probe.py:72-82
            n = self.max_sequence_length or 1000
            pe = torch.zeros(n, d)
            pos = torch.arange(0, n, dtype=torch.float).unsqueeze(1)
            div = torch.exp(torch.arange(0, d, 2).float() * (-torch.log(torch.tensor(10000.0)) / d))
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            self.register_buffer("pos_encoding", pe.unsqueeze(0))
        else:
            self.pos_encoding = None
        self._buf("classifier.weight", self.num_classes, d)
        self._buf("classifier.bias", self.num_classes)

    @torch.no_grad()
    def forward(self, x, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self._combine(self._get_embeddings(x, padding_mask))
        B, T, D = h.shape
        if self.pos_encoding is not None:
            h = h + self.pos_encoding[:, :T]
        if padding_mask is not None and padding_mask.shape[1] != T:   # attention_probe.py:124-125
            padding_mask = None
        if self._stack_ok(D, self.num_heads, 0):
            h = self._stack_forward(h.contiguous(), padding_mask, self.num_heads, self.num_layers, 0)
            return K.dense_f32(K.mean_pool(h), self._p("classifier.weight"), self._p("classifier.bias"))
        for i in range(self.num_layers):
            qkv = K.dense_f32(h, self._p(f"attention_layers.{i}.in_proj_weight"), self._p(f"attention_layers.{i}.in_proj_bias"))
            att = K.mha_f32(qkv, self.num_heads, padding_mask)
            y = K.dense_f32(att, self._p(f"attention_layers.{i}.out_proj.weight"), self._p(f"attention_layers.{i}.out_proj.bias"), resid=h)
            h = K.layernorm(y.reshape(B * T, D), self._p(f"layer_norms.{i}.weight"), self._p(f"layer_norms.{i}.bias"), eps=1e-5,
                            want_f32=True, want_half=False)[0].reshape(B, T, D)
        pooled = K.mean_pool(h)
        return K.dense_f32(pooled, self._p("classifier.weight"), self._p("classifier.bias"))

    def _stack_table(self) -> dict:
        t = {}
        for i in range(self.num_layers):
            for leaf in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"):
                t[f"layers.{i}.self_attn." + leaf.replace("in_proj_", "in_proj.")] = f"attention_layers.{i}.{leaf}"
            for leaf in ("weight", "bias"):
                t[f"layers.{i}.norm1.{leaf}"] = f"layer_norms.{i}.{leaf}"
        return t


class TransformerProbe(_DeviceProbe):
    """transformer_probe.py:17-116: optional LEARNED positions (a Parameter, unlike the attention probe's sinusoids), ``num_layers``
    post-LN ``nn.TransformerEncoderLayer`` s (self attention -> LayerNorm(x + attn) -> Linear / ReLU / Linear of width ``attention_dim``
    -> LayerNorm(x + ff)), mean over the sequence, classifier.  With a key padding mask PyTorch's encoder (eval mode: nested-tensor
    path) returns ZEROS at the padded positions and the reference's mean over all T positions includes them: mirrored."""

    rank = 3

    def __init__(self, base_model, layers, num_classes, device="cuda", feature_mode=False, input_dim=None, aggregation="mean",
                 num_heads: int = 12, attention_dim: int = 768, num_layers: int = 4, dropout_rate: float = 0.1,
                 max_sequence_length: Optional[int] = None, use_positional_encoding: bool = False, target_length=None,
                 freeze_backbone=True) -> None:
        self.num_heads = num_heads
        self.attention_dim = attention_dim
        self.num_layers = num_layers
        self.dropout_rate = dropout_rate
        self.max_sequence_length = max_sequence_length
        self.use_positional_encoding = use_positional_encoding
        super().__init__(base_model, layers, num_classes, device, feature_mode, input_dim, aggregation, target_length, freeze_backbone)

    def build_head(self, d: int) -> None:
        if d % self.num_heads != 0:                                 # transformer_probe.py:58-63: the largest divisor below
            adjusted = min(self.num_heads, d)
            while d % adjusted != 0 and adjusted > 1:
                adjusted -= 1
            self.num_heads = adjusted
        f = self.attention_dim
        for i in range(self.num_layers):
            p = f"transformer.layers.{i}."
            self._buf(p + "self_attn.in_proj_weight", 3 * d, d)
            self._buf(p + "self_attn.in_proj_bias", 3 * d)
            self._buf(p + "self_attn.out_proj.weight", d, d)
            self._buf(p + "self_attn.out_proj.bias", d)
            self._buf(p + "linear1.weight", f, d)
            self._buf(p + "linear1.bias", f)
            self._buf(p + "linear2.weight", d, f)
            self._buf(p + "linear2.bias", d)
            for n in ("norm1", "norm2"):
                self._buf(p + n + ".weight", d)
                self._buf(p + n + ".bias", d)
                self.get_buffer(p + n + ".weight").fill_(1.0)
        if self.use_positional_encoding:
            self.register_buffer("pos_encoding", torch.zeros(1, self.max_sequence_length or 1000, d))
        else:
            self.pos_encoding = None
        self._buf("classifier.weight", self.num_classes, d)
        self._buf("classifier.bias", self.num_classes)

    @torch.no_grad()
    def forward(self, x, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self._combine(self._get_embeddings(x, padding_mask))
        B, T, D = h.shape
        if self.pos_encoding is not None:
            h = h + self.pos_encoding[:, :T]
        if padding_mask is not None and padding_mask.shape[1] != T:   # transformer_probe.py:109-110
            padding_mask = None
        if self._stack_ok(D, self.num_heads, self.attention_dim):
            h = self._stack_forward(h.contiguous(), padding_mask, self.num_heads, self.num_layers, self.attention_dim)
            if padding_mask is not None:
                h = h * (~padding_mask.to(device=h.device, dtype=torch.bool)).unsqueeze(-1).to(h.dtype)
            return K.dense_f32(K.mean_pool(h.contiguous()), self._p("classifier.weight"), self._p("classifier.bias"))
        for i in range(self.num_layers):
            p = f"transformer.layers.{i}."
            qkv = K.dense_f32(h, self._p(p + "self_attn.in_proj_weight"), self._p(p + "self_attn.in_proj_bias"))
            att = K.mha_f32(qkv, self.num_heads, padding_mask)
            y = K.dense_f32(att, self._p(p + "self_attn.out_proj.weight"), self._p(p + "self_attn.out_proj.bias"), resid=h)
            h = K.layernorm(y.reshape(B * T, D), self._p(p + "norm1.weight"), self._p(p + "norm1.bias"), eps=1e-5, want_f32=True, want_half=False)[0]
            f = K.dense_f32(h, self._p(p + "linear1.weight"), self._p(p + "linear1.bias"), act="relu")
            y = K.dense_f32(f, self._p(p + "linear2.weight"), self._p(p + "linear2.bias"), resid=h)
            h = K.layernorm(y, self._p(p + "norm2.weight"), self._p(p + "norm2.bias"), eps=1e-5, want_f32=True, want_half=False)[0].reshape(B, T, D)
        if padding_mask is not None:
            h = h * (~padding_mask.to(device=h.device, dtype=torch.bool)).unsqueeze(-1).to(h.dtype)
        pooled = K.mean_pool(h.contiguous())
        return K.dense_f32(pooled, self._p("classifier.weight"), self._p("classifier.bias"))

    def _stack_table(self) -> dict:
        return {n[len("transformer."):].replace("in_proj_weight", "in_proj.weight").replace("in_proj_bias", "in_proj.bias"): n
                for n, _ in self.named_buffers() if n.startswith("transformer.layers.")}


class LSTMProbe(_DeviceProbe):
    """lstm_probe.py:17-104: optional learned positions, ``nn.LSTM(batch_first=True)`` with ``num_layers`` layers (optionally
    bidirectional) of ``max(lstm_hidden_size, max_sequence_length // 4)`` units, mean over the sequence, classifier.  Per layer and
    direction: one dense product for the input half of the gates, then the whole recurrence in ONE launch (``avexhip_lstm_layer``)."""

    rank = 3

    def __init__(self, base_model, layers, num_classes, device="cuda", feature_mode=False, input_dim=None, aggregation="mean",
                 lstm_hidden_size: int = 256, num_layers: int = 2, bidirectional: bool = False, dropout_rate: float = 0.1,
                 max_sequence_length: Optional[int] = None, use_positional_encoding: bool = False, target_length=None,
                 freeze_backbone=True) -> None:
        self.lstm_hidden_size = lstm_hidden_size
        self.num_layers = num_layers
        self.bidirectional = bidirectional
        self.dropout_rate = dropout_rate
        self.max_sequence_length = max_sequence_length
        self.use_positional_encoding = use_positional_encoding
        super().__init__(base_model, layers, num_classes, device, feature_mode, input_dim, aggregation, target_length, freeze_backbone)

    def build_head(self, d: int) -> None:
        hs = int(max(int((self.max_sequence_length or 4) / 4), self.lstm_hidden_size))      # lstm_probe.py:60
        if hs > 1024:
            raise ValueError(f"LSTMProbe on the device takes at most 1024 hidden units, got {hs}")
        self.hidden = hs
        dirs = 2 if self.bidirectional else 1
        for layer in range(self.num_layers):
            width = d if layer == 0 else hs * dirs
            for suffix in ("", "_reverse")[:dirs]:
                k = f"l{layer}{suffix}"
                self._buf(f"lstm.weight_ih_{k}", 4 * hs, width)
                self._buf(f"lstm.weight_hh_{k}", 4 * hs, hs)
                self._buf(f"lstm.bias_ih_{k}", 4 * hs)
                self._buf(f"lstm.bias_hh_{k}", 4 * hs)
        self._buf("classifier.weight", self.num_classes, hs * dirs)
        self._buf("classifier.bias", self.num_classes)
        if self.use_positional_encoding:
            self.register_buffer("pos_encoding", torch.zeros(1, self.max_sequence_length or 1000, d))
        else:
            self.pos_encoding = None

    @torch.no_grad()
    def forward(self, x, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self._combine(self._get_embeddings(x, padding_mask))        # (the padding mask does not reach the LSTM in the reference either)
        B, T, _ = h.shape
        if self.pos_encoding is not None:
            h = h + self.pos_encoding[:, :T]
        hs, dirs = self.hidden, (2 if self.bidirectional else 1)
        for layer in range(self.num_layers):
            out = torch.empty((B, T, hs * dirs), dtype=torch.float32, device=h.device)
            xgs, whs = [], []
            for suffix in ("", "_reverse")[:dirs]:
                k = f"l{layer}{suffix}"
                xgs.append(K.dense_f32(h.contiguous(), self._p(f"lstm.weight_ih_{k}"), self._p(f"lstm.bias_ih_{k}") + self._p(f"lstm.bias_hh_{k}")))
                whs.append(self._p(f"lstm.weight_hh_{k}").t().contiguous())
            if dirs == 2:      # the two recurrences are independent: one launch, side by side on the chip
                K.lstm_layer_pair(xgs[0], whs[0], xgs[1], whs[1], out)
            else:
                K.lstm_layer(xgs[0], whs[0], out)
            h = out
        pooled = K.mean_pool(h)
        return K.dense_f32(pooled, self._p("classifier.weight"), self._p("classifier.bias"))


PROBES: Dict[str, type] = {"linear": LinearProbe, "mlp": MLPProbe, "attention": AttentionProbe, "transformer": TransformerProbe, "lstm": LSTMProbe}


# ---- the reference's probe registry / factory (models/probes/utils/registry.py:28-93, factory.py:24-186) ---------------------------------
def get_probe_class(name: str) -> Optional[type]:
    return PROBES.get(str(name).lower())


def list_probe_classes() -> List[str]:
    return list(PROBES)


_PROBE_PARAMS = ("hidden_dims", "dropout_rate", "activation", "lstm_hidden_size", "num_layers", "bidirectional", "max_sequence_length",
                 "use_positional_encoding", "num_heads", "attention_dim")      # factory.py:35-46


def build_probe_from_config(probe_config, num_classes: int, device: str, base_model=None, input_dim=None, target_length: Optional[int] = None,
                            **kwargs):
    """``build_probe_from_config`` of the reference (factory.py:56-186) over the device probes: ``probe_config`` is the reference's
    ``ProbeConfig`` (or any object / mapping with its fields: probe_type, target_layers, aggregation, input_processing, freeze_backbone,
    target_length and the probe-specific ones); exactly one of ``base_model`` (online) and ``input_dim`` (offline, feature mode).  Hooks
    are registered on the base model, probe-specific fields that are set are passed on, everything is filtered by the probe class's
    signature -- the same steps, the same errors."""
    import inspect
    from collections.abc import Mapping

    def field(name, default=None):
        return probe_config.get(name, default) if isinstance(probe_config, Mapping) else getattr(probe_config, name, default)

    if base_model is not None and input_dim is not None:
        raise ValueError("Cannot specify both 'base_model' and 'input_dim'. Use 'base_model' for online mode or 'input_dim' for offline mode.")
    if base_model is None and input_dim is None:
        raise ValueError("Must specify either 'base_model' (for online mode) or 'input_dim' (for offline mode).")
    feature_mode = base_model is None
    probe_type = str(field("probe_type")).lower()
    cls = get_probe_class(probe_type)
    if cls is None:
        raise ValueError(f"Probe class '{probe_type}' is not registered. Available classes: {list_probe_classes()}")
    frozen = bool(field("freeze_backbone", True)) if not feature_mode else True
    layers = list(field("target_layers", []) or [])
    if field("input_processing", "pooled") == "sequence" and probe_type not in ("lstm", "attention", "transformer"):
        raise ValueError(f"Sequence input processing is not compatible with {probe_type} probe")
    if not feature_mode and hasattr(base_model, "register_hooks_for_layers"):
        layers = base_model.register_hooks_for_layers(layers)
    init = {"base_model": base_model, "layers": layers, "num_classes": num_classes, "device": device, "feature_mode": feature_mode,
            "input_dim": input_dim, "aggregation": field("aggregation", "mean"),
            "target_length": target_length if target_length is not None else field("target_length"), "freeze_backbone": frozen, **kwargs}
    for name in _PROBE_PARAMS:
        value = field(name)
        if value is not None and value != "":
            init[name] = value
    valid = set(inspect.signature(cls.__init__).parameters)
    return cls(**{k: v for k, v in init.items() if k in valid})
