"""Data-parallel embedding extraction across the GPUs of one node (one process per GPU).

The reference extracts embeddings on a single device (avex/run_evaluate.py:1053);
clips are independent, so the batch is split contiguously over ranks, weights are replicated and
the only exchange is ONE all-gather of the pooled ``[B_local, D]`` embeddings per batch
(RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).  Row i of the result is clip i on
every rank.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from torchrun's environment; returns ``(rank, world_size, local_rank)``.
    A single-process run (no WORLD_SIZE) initialises nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``n % world`` ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather ``[n_local, D]`` shards (possibly ragged) into ``[n_total, D]`` in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    D = local.shape[1]
    if all(hi - lo == width for lo, hi in sizes):
        out = torch.empty((n_total, D), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((width, D), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * width, D), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def extract_embeddings_sharded(embed_fn: Callable[[torch.Tensor], torch.Tensor], wav: torch.Tensor,
                               group=None) -> torch.Tensor:
    """Every rank passes the same ``[B, T]`` batch (or at least its own slice of it); each embeds clips
    ``shard_bounds(B, rank, world)`` with ``embed_fn`` (``[b, T] -> [b, D]``) and the pooled embeddings
    are all-gathered so that every rank returns the full ``[B, D]`` matrix."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return embed_fn(wav)
    lo, hi = shard_bounds(wav.shape[0], dist.get_rank(group), dist.get_world_size(group))
    local = embed_fn(wav[lo:hi]) if hi > lo else None
    if local is None:
        probe = embed_fn(wav[:1])
        local = probe[:0]
    return all_gather_rows(local, wav.shape[0], group)
