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

__all__ = ["extract_embeddings_in_memory", "extract_embeddings_streaming", "write_embedding_metadata"]


def _stage(batch: Dict[str, Any], device: torch.device, stream: Optional["torch.cuda.Stream"], rows: Optional[Tuple[int, int]] = None):
    """Host -> device copy of one Collater batch (on ``stream`` when given: pinned source, non-blocking); ``rows`` = this rank's
    slice of the batch in a sharded run (only those clips cross PCIe)."""
    wav, mask = batch["raw_wav"], batch.get("padding_mask")
    if rows is not None:
        wav = wav[rows[0]:rows[1]]
        mask = mask[rows[0]:rows[1]] if mask is not None else None
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


def _check_same_batches(batch: Optional[Dict[str, Any]], n_batches: Optional[int], tdist: Any, group: Any, device: torch.device) -> None:
    """Sharded extraction assumes every rank iterates the same batches.  Every rank calls this UNCONDITIONALLY before it looks at its first
    batch -- a rank whose dataloader is empty or shorter would otherwise leave the others in a collective it never joins -- with a signature
    of what it is about to iterate: whether it has a first batch, ``len(dataloader)`` when the loader has one, and the first batch's clip
    count, length and checksums of audio and labels.  (A sampler that diverges only AFTER the first batch at equal length is not detected:
    the check is a guard against the common mis-configuration, a per-rank sampler, not a proof.)"""
    sig = [1.0 if batch is not None else 0.0, float(n_batches) if n_batches is not None else -1.0, 0.0, 0.0, 0.0, 0.0]
    if batch is not None:
        wav = batch["raw_wav"]
        lab = batch.get("label")
        sig[2:5] = [float(wav.shape[0]), float(wav.shape[-1]), float(wav.detach().double().abs().sum().item())]
        if torch.is_tensor(lab):
            sig[5] = float(lab.detach().double().sum().item())
    backend = tdist.get_backend(group)
    t = torch.tensor(sig, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    lo, hi = t.clone(), t.clone()
    tdist.all_reduce(lo, op=tdist.ReduceOp.MIN, group=group)
    tdist.all_reduce(hi, op=tdist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError("extract_embeddings_in_memory(sharded=True): the ranks do not iterate the same batches ([has a first batch, len(dataloader), "
                           f"clips, samples, |audio| sum, label sum]: min {lo.tolist()} max {hi.tolist()}).  Sharded extraction splits every batch over "
                           "the ranks; with a DistributedSampler (a different batch per rank) call it with sharded=False.")


def extract_embeddings_in_memory(model: Any, dataloader: Iterable[Dict[str, Any]], target_layers: List[Any],
                                 device: Any, aggregation: str = "mean", disable_tqdm: bool = True,
                                 disable_layerdrop: Optional[bool] = None, prefetch: Optional[bool] = None,
                                 sharded: bool = False, group: Any = None, batch_invariant: Optional[bool] = None
                                 ) -> Tuple[Dict[str, torch.Tensor], torch.Tensor, List[tuple]]:
    """Run ``model.extract_embeddings`` over every batch of ``dataloader`` and stack the results on the CPU
    (reference: embedding_utils.py:26-144; ``disable_tqdm`` is accepted for signature compatibility, no progress bar here).

    ``sharded=True`` (opt-in; needs ``torch.distributed`` with more than one rank): every rank iterates the SAME batches, embeds clips
    ``dist.shard_bounds(B, rank, world)`` of each and the embeddings are all-gathered in clip order (``dist.PipelinedGather``: batch
    n's exchange under batch n + 1's forward), so every rank returns what a single device would (the
    reference's loop is single-device, run_evaluate.py:1053).  It is NOT the default: a multi-rank caller with a
    ``DistributedSampler`` -- the usual set-up -- hands every rank DIFFERENT batches, and sharding those would gather rows of unrelated
    clips beside the local labels without any visible failure.  The first batch's signature (clip count, samples, a checksum of the
    audio and of the labels), whether there IS a first batch and ``len(dataloader)`` are compared across ranks before anything else and a
    mismatch raises ``RuntimeError`` on every rank.

    ``batch_invariant`` (``True`` / ``False``; default: leave the model as it is): make the model's HIP handle give every clip the same
    bits whatever batch it arrives in -- the loop's last, partial batch then rounds exactly like the full ones (``kernels.residual_code``;
    the reference's fp32 path is batch-independent, beats_model.py:279-429).  The handle is rebuilt if the setting changes."""
    from . import dist as adist
    import torch.distributed as tdist
    if batch_invariant is not None and hasattr(model, "batch_invariant") and bool(model.batch_invariant) != bool(batch_invariant):
        model.batch_invariant = bool(batch_invariant)
        model._weights_dirty = True      # the next forward builds its handle again, with the other policy
    world = tdist.get_world_size(group) if tdist.is_available() and tdist.is_initialized() else 1
    rank = tdist.get_rank(group) if world > 1 else 0
    sharded = bool(sharded) and world > 1
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
            gathers: Dict[int, Any] = {}                        # position in the output list -> PipelinedGather
            if sharded:
                try:
                    n_batches: Optional[int] = len(dataloader)      # type: ignore[arg-type]
                except TypeError:
                    n_batches = None
                _check_same_batches(nxt, n_batches, tdist, group, device)

            def my_rows(b: Dict[str, Any]) -> Optional[Tuple[int, int]]:
                return adist.shard_bounds(int(b["raw_wav"].shape[0]), rank, world) if sharded else None

            def exchange(parts: List[Tuple[str, torch.Tensor]], n_total: int) -> List[Tuple[str, torch.Tensor]]:
                """Sharded run: start this batch's all-gathers, hand back the PREVIOUS batch's finished ones (clip order)."""
                done = []
                for i, (name, t) in enumerate(parts):
                    if i not in gathers:
                        gathers[i] = [adist.PipelinedGather(group), None]
                    pg = gathers[i]
                    prev = pg[0].push(t, n_total)
                    if prev is not None:       # (a host tensor aliases the gather's buffer, which the push after next overwrites)
                        done.append((pg[1], prev if prev.is_cuda else prev.clone()))
                    pg[1] = name
                return done

            staged = _stage(nxt, device, copy_stream, my_rows(nxt)) if nxt is not None else None
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
                    staged = _stage(nxt, device, copy_stream, my_rows(nxt))          # overlaps the forward below
                empty_shard = sharded and wav.shape[0] == 0                          # fewer clips than ranks: embed one clip for the shapes, keep no row
                if empty_shard:
                    wav_in = batch["raw_wav"][:1].to(device)
                    mask_in = batch["padding_mask"][:1].to(device) if batch.get("padding_mask") is not None else None
                else:
                    wav_in, mask_in = wav, mask
                if mask_in is None:
                    emb = model.extract_embeddings(wav_in, aggregation=aggregation)
                else:
                    emb = model.extract_embeddings({"raw_wav": wav_in, "padding_mask": mask_in}, aggregation=aggregation)
                parts: List[Tuple[str, torch.Tensor]] = []
                if isinstance(emb, list):                                            # embedding_utils.py:98-104
                    parts = [(resolved_layers[i] if i < len(resolved_layers) else f"layer_{i}", e) for i, e in enumerate(emb)]
                elif isinstance(emb, dict):
                    parts = list(emb.items())
                else:
                    parts = [(resolved_layers[0] if resolved_layers else "embeddings", emb)]
                if empty_shard:
                    parts = [(n, t[:0]) for n, t in parts]
                if sharded:
                    parts = exchange(parts, int(batch["raw_wav"].shape[0]))          # the previous batch's rows, gathered
                outs: List[Tuple[str, torch.Tensor]] = []
                for name, t in parts:
                    to_host(name, t, outs)
                if copy_stream is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                    pending.append((ev, outs))
                else:
                    for name, t in outs:
                        layer_embeds.setdefault(name, []).append(t)
                labels.append(batch["label"].cpu())
            if sharded:                                                              # the last batch's exchange
                outs = []
                for i in sorted(gathers):
                    pg, name = gathers[i]
                    last = pg.flush()
                    if last is not None:
                        to_host(name, last if last.is_cuda else last.clone(), outs)
                if copy_stream is not None and outs:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                    pending.append((ev, outs))
                else:
                    for name, t in outs:
                        layer_embeds.setdefault(name, []).append(t)
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


def write_embedding_metadata(h5f: Any, *, aggregation: str, layer_names: List[str], embedding_dims: List[tuple], multi_layer: bool) -> None:
    """The cache attributes a reader needs to interpret the stored shapes (reference: embedding_utils.py:147-161, same keys and
    value forms).  ``h5f`` is an ``h5py.File`` or anything with a mapping ``.attrs``."""
    h5f.attrs["embedding_aggregation"] = aggregation
    h5f.attrs["aggregation"] = aggregation
    h5f.attrs["stored_embedding_rank"] = [len(tuple(dim)) for dim in embedding_dims]
    h5f.attrs["layer_names"] = layer_names
    h5f.attrs["embedding_dims"] = [str(tuple(dim)) for dim in embedding_dims]
    h5f.attrs["multi_layer"] = multi_layer


def extract_embeddings_streaming(model: Any, dataloader: Any, target_layers: List[Any], device: Any, save_path: Any,
                                 chunk_size: int = 1000, compression: Optional[str] = "gzip", compression_level: int = 4,
                                 aggregation: str = "mean", disable_layerdrop: Optional[bool] = None, h5_module: Any = None,
                                 prefetch: Optional[bool] = None) -> List[tuple]:
    """Extraction straight into an HDF5 cache (reference: ``_extract_embeddings_streaming`` + ``_create_and_fill_h5_datasets_hybrid``,
    embedding_utils.py:164-346, 349-822): one float32 dataset ``embeddings_<layer>`` per hooked layer, shape ``(N, *dim)`` in chunks of
    ``(chunk_size, *dim)``, an int64 ``labels`` dataset, the attributes of ``write_embedding_metadata`` plus ``num_labels`` /
    ``label_shape`` / ``extraction_complete`` / ``skipped_batches``.  Returns the per-layer embedding dims like the reference.

    The device side is the loop of ``extract_embeddings_in_memory`` (overlapped PCIe copies); rows are written as each batch's
    device-to-host copy completes, so host memory holds a few batches, not the data set.  HDF5 itself is a storage concern outside
    the accelerated path: ``h5py`` is imported here, at call time (``h5_module`` lets a caller or a test inject a compatible
    module); without it this raises ``ImportError`` and nothing else in the package is affected."""
    import os

    import numpy as np
    if h5_module is None:
        try:
            import h5py as h5_module                     # type: ignore[no-redef]
        except ImportError as e:                         # pragma: no cover - environment dependent
            raise ImportError("extract_embeddings_streaming writes an HDF5 cache and needs h5py") from e
    total = len(dataloader.dataset)                       # embedding_utils.py:240
    if chunk_size <= 0:
        raise ValueError(f"Invalid chunk size: {chunk_size}. Must be positive.")
    parent = os.path.dirname(os.fspath(save_path))
    if parent:
        os.makedirs(parent, exist_ok=True)
    kwargs: Dict[str, Any] = {}
    if compression and str(compression).lower() not in {"none", "null", "false"}:         # embedding_utils.py:416-421
        kwargs["compression"] = compression
        if str(compression).lower() != "lzf":
            kwargs["compression_opts"] = int(compression_level)

    state: Dict[str, Any] = {"dsets": None, "labels": None, "dims": None, "row": 0, "label_shape": None}

    def sink(h5f: Any, names: List[str], embs: List[torch.Tensor], labels: torch.Tensor) -> None:
        n = int(embs[0].shape[0])
        if state["dsets"] is None:                        # first batch fixes names, dims and the label layout
            dims = [tuple(e.shape[1:]) for e in embs]
            per_sample = int(sum(int(np.prod(d)) if len(d) else 1 for d in dims))
            cs = min(chunk_size, max(1, int(4 * 1024 ** 3 / max(1, per_sample * 4))), max(1, total))      # HDF5's 4 GB chunk limit (:276-291)
            state["dsets"] = {nm: h5f.create_dataset(f"embeddings_{nm}", shape=(total,) + d, maxshape=(None,) + d, dtype=np.float32,
                                                     chunks=(cs,) + d, **kwargs) for nm, d in zip(names, dims)}
            ls = tuple(labels.shape[1:]) if labels.dim() > 1 else ()
            np_dt = np.dtype(str(labels.dtype).replace("torch.", "")) if str(labels.dtype) != "torch.bool" else np.bool_
            state["labels"] = h5f.create_dataset("labels", shape=(total,) + ls, maxshape=(None,) + ls, dtype=np_dt, chunks=(cs,) + ls, **kwargs)
            state["dims"], state["label_shape"] = dims, ls
        r = state["row"]
        for nm, e in zip(names, embs):
            state["dsets"][nm][r:r + n] = e.numpy().astype(np.float32)
        state["labels"][r:r + n] = labels.numpy().astype(np.int64)                           # embedding_utils.py:548
        state["row"] = r + n

    device = torch.device(device)
    if prefetch is None:
        prefetch = device.type == "cuda"
    prefetch = bool(prefetch) and device.type == "cuda" and torch.cuda.is_available()
    original_disable_layerdrop = None
    if disable_layerdrop is not None and hasattr(model, "disable_layerdrop"):
        original_disable_layerdrop = model.disable_layerdrop
        model.disable_layerdrop = disable_layerdrop
    copy_stream = torch.cuda.Stream(device=device) if prefetch else None
    try:
        with torch.no_grad(), h5_module.File(os.fspath(save_path), "w") as h5f:
            resolved_layers = model.register_hooks_for_layers(target_layers)
            pending: List[Tuple[Any, List[str], List[torch.Tensor], torch.Tensor]] = []

            def drain(keep: int) -> None:
                while len(pending) > keep:
                    ev, names, embs, lab = pending.pop(0)
                    if ev is not None:
                        ev.synchronize()
                    sink(h5f, names, embs, lab)

            it = iter(dataloader)
            nxt = next(it, None)
            staged = _stage(nxt, device, copy_stream) if nxt is not None else None
            while nxt is not None:
                batch, (wav, mask, ready) = nxt, staged
                nxt = next(it, None)
                if ready is not None:
                    cur = torch.cuda.current_stream(device)
                    cur.wait_event(ready)
                    wav.record_stream(cur)
                    if mask is not None:
                        mask.record_stream(cur)
                if nxt is not None:
                    staged = _stage(nxt, device, copy_stream)
                emb = model.extract_embeddings(wav if mask is None else {"raw_wav": wav, "padding_mask": mask}, aggregation=aggregation)
                if isinstance(emb, list):
                    names = [resolved_layers[i] if i < len(resolved_layers) else f"layer_{i}" for i in range(len(emb))]
                    embs = list(emb)
                elif isinstance(emb, dict):
                    names, embs = list(emb.keys()), list(emb.values())
                else:
                    names, embs = [resolved_layers[0] if resolved_layers else "embeddings"], [emb]
                host: List[torch.Tensor] = []
                for t in embs:
                    if copy_stream is not None and t.is_cuda:
                        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                        h.copy_(t, non_blocking=True)
                        host.append(h)
                    else:
                        host.append(t.cpu())
                ev = None
                if copy_stream is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(device))
                pending.append((ev, names, host, batch["label"].cpu()))
                drain(2)                                   # at most two batches of embeddings wait on the host
            drain(0)
            if state["dsets"] is None:
                raise ValueError("No data processed. Check if dataloader is empty or has invalid batches.")
            ls = state["label_shape"]
            if ls:                                         # embedding_utils.py:773-787
                if len(ls) == 1 and ls[0] > 1:
                    h5f.attrs["num_labels"] = ls[0]
                else:
                    h5f.attrs["num_labels"] = -1
                    h5f.attrs["label_shape"] = ls
            else:
                h5f.attrs["num_labels"] = len(np.unique(np.asarray(state["labels"][:])))
            write_embedding_metadata(h5f, aggregation=aggregation, layer_names=list(state["dsets"].keys()),
                                     embedding_dims=state["dims"], multi_layer=True)
            if state["row"] != total:
                logger.warning("Expected to process %d samples, but processed %d", total, state["row"])
            h5f.attrs["extraction_complete"] = state["row"] == total
            h5f.attrs["skipped_batches"] = 0
        return [tuple(d) for d in state["dims"]]
    finally:
        if original_disable_layerdrop is not None and hasattr(model, "disable_layerdrop"):
            model.disable_layerdrop = original_disable_layerdrop
        model.deregister_all_hooks()
