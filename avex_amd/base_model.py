"""``ModelBase``: the plugin contract every registered model class implements.

Behavioural mirror of the reference's base class (avex/models/base_model.py:19-457): same public
methods, argument meaning, attribute names (``_hooks``, ``_hook_layers``, ``_hook_outputs``,
``_layer_names``, ``audio_processor``, ``device``) and exceptions, so probes and evaluation drivers
written against the reference work unchanged on a model from this package.

Hooks are ordinary ``torch`` forward hooks on named sub-modules.  Model classes whose forward is a
fused HIP path (``avex_amd.beats_model.Model``) never *call* those sub-modules; they deliver each
tapped layer's raw output with :meth:`ModelBase._fire_forward_hooks`, which invokes whatever hooks
are registered on the module exactly like ``nn.Module.__call__`` would.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Mapping, Optional, Union

import torch
import torch.nn as nn

from .configs import AudioConfig

logger = logging.getLogger(__name__)


def coerce_audio_config(audio_config: Any) -> Optional[AudioConfig]:
    """This package's ``AudioConfig`` from whatever a caller hands to a model constructor.

    The reference's factory passes ITS OWN pydantic ``avex.configs.AudioConfig`` (factory.py:132-143), YAML loaders pass
    mappings, and callers of this package pass ``avex_amd.configs.AudioConfig``: all three carry the same fields.  Foreign
    objects are read through ``model_dump()`` (pydantic v2), ``dict()`` (v1) or their attributes; fields this package does
    not know are dropped (the schema here is the reference's field list), a missing field keeps its default."""
    if audio_config is None or isinstance(audio_config, AudioConfig):
        return audio_config
    if isinstance(audio_config, Mapping):
        return AudioConfig(**audio_config)          # a mapping is validated strictly (unknown keys are an error, as in the reference)
    fields = AudioConfig.model_fields
    for getter in ("model_dump", "dict"):
        fn = getattr(audio_config, getter, None)
        if callable(fn):
            data = fn()
            if isinstance(data, Mapping):
                return AudioConfig(**{k: v for k, v in data.items() if k in fields})
    data = {k: getattr(audio_config, k) for k in fields if hasattr(audio_config, k)}
    if not data:
        raise TypeError(f"audio_config must be an AudioConfig, a mapping or an object with its fields, got {type(audio_config).__name__}")
    return AudioConfig(**data)


class AudioProcessor:
    """Waveform pre-processing selected by an ``AudioConfig`` (reference: data/audio_utils.py:76-179).

    ``"raw"`` is the identity (what every BEATs spec uses).  ``"spectrogram"`` / ``"mel_spectrogram"`` (the EfficientNet
    frontend: ``torch.stft`` -> power -> ``MelScale`` -> log + per-clip min-max) run on the GPU through
    ``avexhip_melspec_forward`` (fp32-MFMA DFT, ``avex_amd/csrc/melspec.hip``); like the reference, the result comes back on
    the device of the input.  There is no CPU implementation: without a GPU these representations raise ``AvexHipError``.
    """

    def __init__(self, cfg: AudioConfig) -> None:
        self.cfg = cfg
        self.sr = cfg.sample_rate
        self.n_fft = cfg.n_fft
        self.hop_length = cfg.hop_length or self.n_fft // 4
        self.win_length = cfg.win_length or self.n_fft
        self.window_type = cfg.window
        self.n_mels = cfg.n_mels
        self.representation = cfg.representation
        self.normalize = cfg.normalize
        self.target_length_seconds = cfg.target_length_seconds
        self.window_selection = cfg.window_selection
        self.center = cfg.center
        self._plan = None

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        if waveform.dim() == 1:
            waveform = waveform.unsqueeze(0)
        if self.representation == "raw":
            return waveform
        if self.representation not in ("spectrogram", "mel_spectrogram"):
            raise ValueError(f"Unknown representation: {self.representation}")
        from . import kernels                     # the HIP library is only needed for the spectrogram representations
        if self._plan is None:
            if self.window_type not in ("hann", "hamming"):
                raise ValueError(f"Unknown window type: {self.window_type}")
            self._plan = kernels.MelspecPlan(n_fft=self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                                             window=self.window_type, n_mels=self.n_mels, sample_rate=self.sr,
                                             mel=self.representation == "mel_spectrogram", center=self.center, normalize=self.normalize)
        original = waveform.device
        x = waveform if waveform.is_cuda else waveform.to(torch.device("cuda", torch.cuda.current_device()))
        return self._plan(x.to(torch.float32)).to(original)


class ModelBase(nn.Module):
    """Hook registry + generic ``extract_embeddings`` + ``process_audio``."""

    def __init__(self, device: str, audio_config: Optional[Union[AudioConfig, Dict[str, Any]]] = None) -> None:
        # nn.Module's initialiser by name, not super(): the reference-side binding lists the
        # reference's own ModelBase as a second base so that its isinstance checks hold (registry.py:695), and that class's
        # __init__(device, audio_config) must not run a second time behind this one in the MRO
        nn.Module.__init__(self)
        self.device = device          # a plain str, deliberately NOT updated by .to() (reference behaviour)
        audio_config = coerce_audio_config(audio_config)
        self.audio_processor = AudioProcessor(audio_config) if audio_config else None
        self._hooks: Dict[str, torch.utils.hooks.RemovableHandle] = {}
        self._hook_outputs: Dict[str, torch.Tensor] = {}
        self._layer_names: List[str] = []
        self._hook_layers: List[str] = []

    # ------------------------------------------------------------------ layer discovery
    def _discover_embedding_layers(self) -> None:
        """Default: every ``nn.Linear`` in ``named_modules()`` order; subclasses narrow this."""
        if not self._layer_names:
            self._layer_names = [n for n, m in self.named_modules() if isinstance(m, nn.Linear)]

    def get_model_layers(self) -> list[str]:
        self._discover_embedding_layers()
        return list(self._layer_names)

    def get_model_layer_map(self) -> dict[int, str]:
        return dict(enumerate(self.get_model_layers()))

    def _get_last_non_classification_layer(self) -> Optional[str]:
        if not self._layer_names:
            return None
        for name in reversed(self._layer_names):
            low = name.lower()
            if "classifier" in low or "head" in low:
                continue
            return name
        return self._layer_names[-1]

    def _get_all_linear_layers(self) -> List[str]:
        return self._layer_names

    # ------------------------------------------------------------------ hooks
    def _create_hook_fn(self, layer_name: str):
        def hook_fn(module: nn.Module, inputs, output) -> None:
            # stored without detach so gradients can flow for torch-forward models
            if isinstance(output, dict):
                output = output["x"]
            elif isinstance(output, tuple):
                output = output[0]
            self._hook_outputs[layer_name] = output
        return hook_fn

    def register_hooks_for_layers(self, target_layers: List[Union[str, int]]) -> List[str]:
        """Resolve selectors (names, 0-based indices incl. negatives, ``"all"``, ``"last_layer"``),
        replace any existing hooks and return the resolved names in registration order."""
        self._discover_embedding_layers()
        names: List[str] = []
        for sel in target_layers:
            if isinstance(sel, bool):
                raise TypeError("target_layers entries must be str or int (bool is not allowed).")
            if isinstance(sel, int):
                n = len(self._layer_names)
                if not -n <= sel < n:
                    raise ValueError(f"Layer index {sel} is out of range for {n} layers "
                                     f"(valid indices: 0..{n - 1} and negative indices like -1).")
                names.append(self._layer_names[sel])
            else:
                names.append(sel)
        if "all" in names:
            names = [n for n in names if n != "all"] + list(self._layer_names)
        if "last_layer" in names:
            last = self._get_last_non_classification_layer()
            if not last:
                raise ValueError("No layers available for 'last_layer'")
            names = [last if n == "last_layer" else n for n in names]
        resolved = list(dict.fromkeys(names))          # order-preserving dedup

        self.deregister_all_hooks()
        self._hook_layers = resolved
        for name in resolved:
            try:
                module = self.get_submodule(name)
            except AttributeError as err:
                raise ValueError(f"Layer '{name}' not found in model") from err
            self._hooks[name] = module.register_forward_hook(self._create_hook_fn(name))
        return resolved

    def ensure_hooks_registered(self) -> None:
        if self._hooks or not self._hook_layers:
            return
        self.register_hooks_for_layers(self._hook_layers)

    def deregister_all_hooks(self) -> None:
        for h in self._hooks.values():
            h.remove()
        self._hooks.clear()
        self._hook_outputs.clear()          # _hook_layers is kept: it records what should be hooked

    def _clear_hook_outputs(self) -> None:
        self._hook_outputs.clear()

    def _cleanup_hooks(self) -> None:
        self.deregister_all_hooks()

    def __del__(self) -> None:
        try:
            self._cleanup_hooks()
        except Exception:  # noqa: BLE001
            pass

    @staticmethod
    def _fire_forward_hooks(module: nn.Module, output: torch.Tensor) -> None:
        """Deliver ``output`` to every forward hook registered on ``module`` (fused-path models)."""
        for hook in list(module._forward_hooks.values()):
            hook(module, (), output)

    # ------------------------------------------------------------------ audio
    def process_audio(self, x: torch.Tensor) -> torch.Tensor:
        if x is None:
            raise ValueError("Input tensor cannot be None")
        if self.audio_processor is not None:
            x = self.audio_processor(x)
        return x.to(next(self.parameters()).device)

    def enable_gradient_checkpointing(self) -> None:
        raise NotImplementedError(f"{self.__class__.__name__} does not support gradient checkpointing.")

    def batch_inference(self, batched_samples) -> torch.Tensor:
        outs: List[torch.Tensor] = []
        for batch in batched_samples:
            emb = self(self.process_audio(batch))
            outs.append(emb.unsqueeze(0) if emb.dim() == 1 else emb)
        return torch.cat(outs, dim=0)

    # ------------------------------------------------------------------ embeddings
    @staticmethod
    def _aggregate(embeddings: List[torch.Tensor], aggregation: str) -> Union[torch.Tensor, List[torch.Tensor]]:
        if aggregation == "none":
            return embeddings[0] if len(embeddings) == 1 else embeddings
        out: List[torch.Tensor] = []
        for e in embeddings:
            if e.dim() == 3:
                if aggregation == "mean":
                    e = e.mean(dim=1)
                elif aggregation == "max":
                    e = e.max(dim=1)[0]
                elif aggregation == "cls_token":
                    e = e[:, 0, :]
                else:
                    raise ValueError(f"Unsupported aggregation method: {aggregation}")
            elif e.dim() != 2:
                raise ValueError(f"Unexpected embedding dimension: {e.dim()}. Expected 2 or 3.")
            out.append(e)
        return out[0] if len(out) == 1 else torch.cat(out, dim=1)

    def extract_embeddings(self, x: Union[torch.Tensor, Dict[str, torch.Tensor]], *,
                           padding_mask: Optional[torch.Tensor] = None, aggregation: str = "none"
                           ) -> Union[torch.Tensor, List[torch.Tensor]]:
        self._clear_hook_outputs()
        self.ensure_hooks_registered()
        if not self._hooks:
            raise ValueError("No hooks registered. Call register_hooks_for_layers() first.")
        try:
            if isinstance(x, dict):
                wav, mask = x["raw_wav"], x.get("padding_mask")
            else:
                wav, mask = x, padding_mask
            batch = wav.shape[0]
            self.forward(wav, mask)
            order = self._hook_layers or list(self._hook_outputs)
            missing = [n for n in order if n not in self._hook_outputs]
            if missing and not self._hook_outputs:
                raise ValueError(f"No layers found matching: {missing}")
            if missing:
                raise ValueError("Some requested layers did not produce hook outputs: "
                                 f"{missing}. Available outputs: {list(self._hook_outputs)}")
            embs = [self._hook_outputs[n] for n in order]
            if not embs:
                raise ValueError(f"No layers found matching: {list(self._hook_outputs)}")
            embs = [e if e.shape[0] == batch else e.transpose(0, 1) for e in embs]   # (T,B,E) taps -> batch first
            return self._aggregate(embs, aggregation)
        finally:
            self._clear_hook_outputs()
