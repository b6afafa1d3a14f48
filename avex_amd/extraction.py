"""In-memory embedding extraction loop: the harness counterpart of the hot path.

Mirrors ``_extract_embeddings_in_memory`` of the reference (avex/evaluation/embedding_utils.py:26-144): same arguments, same
return triple ``(embeddings: dict layer -> CPU tensor, labels: CPU tensor, embedding_dims)``, same batch schema (the reference's
``Collater`` yields ``{"raw_wav", "padding_mask", "label"}``, avex/data/dataset.py:393-399), same error and clean-up behaviour
(``ValueError`` when nothing was processed, hooks deregistered and ``disable_layerdrop`` restored in ``finally``).

MI355X-first difference: with ``prefetch=True`` (default on a GPU) batch i+1 is staged into pinned host memory and copied
to HBM on a side stream while batch i runs, and the device-to-host copy of batch i's embeddings is issued asynchronously
into pinned buffers, so the PCIe legs overlap the encoder instead of serialising with it (the reference does
``.to(device)`` / ``.cpu()`` synchronously around every batch).  The numbers that come out are the same.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Iterable, List, Optional, Tuple

import torch

logger = logging.getLogger(__name__)

__all__ = ["extract_embeddings_in_memory"]


def _stage(batch: Dict[str, Any], device: torch.device, stream: Optional["torch.cuda.Stream"]):
    """Host -> device copy of one Collater batch (on ``stream`` when given: pinned source, non-blocking)."""
    wav, mask = batch["raw_wav"], batch.get("padding_mask")
    if stream is None:
        return wav.to(device), (mask.to(device) if mask is not None else None), None
    with torch.cuda.stream(stream):
        w = (wav if wav.is_pinned() or wav.is_cuda else wav.pin_memory()).to(device, non_blocking=True)
        m = None
        if mask is not None:
            m = (mask if mask.is_pinned() or mask.is_cuda else mask.pin_memory()).to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
    return w, m, ev


def extract_embeddings_in_memory(model: Any, dataloader: Iterable[Dict[str, Any]], target_layers: List[Any],
                                 device: Any, aggregation: str = "mean", disable_tqdm: bool = True,
                                 disable_layerdrop: Optional[bool] = None, prefetch: Optional[bool] = None
                                 ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, List[tuple]]:
    """Run ``model.extract_embeddings`` over every batch of ``dataloader`` and stack the results on the CPU
    (reference: embedding_utils.py:26-144; ``disable_tqdm`` is accepted for signature compatibility, no progress bar here)."""
    device = torch.device(device)
    if prefetch is None:
        prefetch = device.type == "cuda"
    prefetch = bool(prefetch) and device.type == "cuda" and torch.cuda.is_available()
    original_disable_layerdrop = None
    if disable_layerdrop is not None and hasattr(model, "disable_layerdrop"):          # embedding_utils.py:49-52
        original_disable_layerdrop = model.disable_layerdrop
        model.disable_layerdrop = disable_layerdrop

    layer_embeds: Dict[str, List[torch.Tensor]] = {}
    labels: List[torch.Tensor] = []
    pending: List[Tuple["torch.cuda.Event", List[Tuple[str, torch.Tensor]]]] = []
    copy_stream = torch.cuda.Stream(device=device) if prefetch else None

    def to_host(name: str, t: torch.Tensor, outs: List[Tuple[str, torch.Tensor]]) -> None:
        if copy_stream is None or not t.is_cuda:
            outs.append((name, t.cpu()))
            return
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        outs.append((name, host))

    try:
        with torch.no_grad():
            resolved_layers = model.register_hooks_for_layers(target_layers)         # outside the loop, like the reference
            it = iter(dataloader)
            nxt = next(it, None)
            staged = _stage(nxt, device, copy_stream) if nxt is not None else None
            while nxt is not None:
                batch, (wav, mask, ready) = nxt, staged
                nxt = next(it, None)
                if ready is not None:
                    cur = torch.cuda.current_stream(device)
                    cur.wait_event(ready)
                    # the staged tensors were allocated on copy_stream's pool but are read by kernels on the compute stream: tell the
                    # allocator, or dropping them at the next iteration hands the block to a later H2D copy that does not wait for
                    # this batch's forward (silently overwritten audio when the host runs ahead)
                    wav.record_stream(cur)
                    if mask is not None:
                        mask.record_stream(cur)
                if nxt is not None:
                    staged = _stage(nxt, device, copy_stream)                        # overlaps the forward below
                if mask is None:
                    emb = model.extract_embeddings(wav, aggregation=aggregation)
                else:
                    emb = model.extract_embeddings({"raw_wav": wav, "padding_mask": mask}, aggregation=aggregation)
                outs: List[Tuple[str, torch.Tensor]] = []
                if isinstance(emb, list):                                            # embedding_utils.py:98-104
                    for i, layer_emb in enumerate(emb):
                        to_host(resolved_layers[i] if i < len(resolved_layers) else f"layer_{i}", layer_emb, outs)
                elif isinstance(emb, dict):
                    for layer_name, layer_emb in emb.items():
                        to_host(layer_name, layer_emb, outs)
                else:
                    to_host(resolved_layers[0] if resolved_layers else "embeddings", emb, outs)
                if copy_stream is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                    pending.append((ev, outs))
                else:
                    for name, t in outs:
                        layer_embeds.setdefault(name, []).append(t)
                labels.append(batch["label"].cpu())
            for ev, outs in pending:
                ev.synchronize()
                for name, t in outs:
                    layer_embeds.setdefault(name, []).append(t)

        if not labels:
            raise ValueError("No data processed. Check if dataloader is empty or has invalid batches.")
        final_embeddings: Dict[str, torch.Tensor] = {}
        embedding_dims: List[tuple] = []
        for layer_name, layer_tensors in layer_embeds.items():
            final_embeddings[layer_name] = torch.cat(layer_tensors)
            embedding_dims.append(tuple(final_embeddings[layer_name].shape[1:]))
        return final_embeddings, torch.cat(labels), embedding_dims
    finally:
        if original_disable_layerdrop is not None and hasattr(model, "disable_layerdrop"):
            model.disable_layerdrop = original_disable_layerdrop
        model.deregister_all_hooks()                                                 # embedding_utils.py:143-144
