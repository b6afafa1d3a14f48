"""Data-parallel embedding extraction across the GPUs of one node (one process per GPU).

The reference extracts embeddings on a single device (avex/run_evaluate.py:1053);
clips are independent, so the batch is split contiguously over ranks, weights are replicated and
the only exchange is ONE all-gather of the pooled ``[B_local, D]`` embeddings per batch
(RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).  Row i of the result is clip i on
every rank.
"""
from __future__ import annotations

import os
from typing import Any, Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


_WORK_GROUP = None      # the group the gathers run on by default: every rank, long collective timeout (init_distributed)
_WORK_WORLD = None      # the default process group _WORK_GROUP was made in: a group of a destroyed world must never be used again


def work_group():
    """The process group ``init_distributed`` made for the data path (``None`` = the default group: single process, initialised elsewhere,
    or torch.distributed was destroyed and re-initialised since -- a group of the old world is dropped, not used)."""
    global _WORK_GROUP, _WORK_WORLD
    if _WORK_GROUP is not None:
        alive = dist.is_initialized() and dist.distributed_c10d._get_default_group() is _WORK_WORLD
        if not alive:
            _WORK_GROUP = _WORK_WORLD = None
    return _WORK_GROUP


def shutdown() -> None:
    """Destroy what ``init_distributed`` made (the working group, then the default group) and forget it, so that a later
    ``init_distributed`` -- or an initialisation done elsewhere -- starts clean."""
    global _WORK_GROUP, _WORK_WORLD
    if dist.is_initialized():
        if work_group() is not None:
            dist.destroy_process_group(_WORK_GROUP)
        dist.destroy_process_group()
    _WORK_GROUP = _WORK_WORLD = None


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from torchrun's environment; returns ``(rank, world_size, local_rank)``.
    A single-process run (no WORLD_SIZE) initialises nothing."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and dist.is_initialized() and work_group() is None:
        # initialised elsewhere (or re-initialised after a destroy): the data path still gets its own group with the long collective
        # timeout, behind a barrier on the default group
        import datetime
        dist.barrier()
        global _WORK_GROUP, _WORK_WORLD
        _WORK_GROUP = dist.new_group(ranks=list(range(world)), backend=dist.get_backend(),
                                     timeout=datetime.timedelta(seconds=int(os.environ.get("AVEX_AMD_DIST_COLLECTIVE_TIMEOUT_S", "1800"))))
        _WORK_WORLD = dist.distributed_c10d._get_default_group()
    if world > 1 and not dist.is_initialized():
        # dmabuf IPC only on these hosts (RCCL across processes).  ROCr reads the flag when the runtime starts, so it goes in BEFORE the first
        # call that initialises HIP in this process -- torch.cuda.is_available() below is one.  (A process that touched the GPU earlier must
        # have it exported by its launcher: bench.py's launcher does.)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        import datetime
        # Two timeouts.  The rendezvous and the first barrier get a short one: a rank that never arrives fails the job instead of hanging it.
        # The working group gets a long one (torch's own default is 10 minutes): its timeout applies to EVERY collective, and a rank may
        # legitimately spend minutes between two gathers (a first-use build of the extension, a slow dataloader).
        t_rdv = datetime.timedelta(seconds=int(os.environ.get("AVEX_AMD_DIST_TIMEOUT_S", "300")))
        t_run = datetime.timedelta(seconds=int(os.environ.get("AVEX_AMD_DIST_COLLECTIVE_TIMEOUT_S", "1800")))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank), timeout=t_rdv)
        else:
            dist.init_process_group(backend=backend, timeout=t_rdv)
        dist.barrier()                                       # everyone is here (under the short timeout)
        _WORK_GROUP = dist.new_group(ranks=list(range(world)), timeout=t_run, backend=backend)
        _WORK_WORLD = dist.distributed_c10d._get_default_group()
    return rank, world, local_rank


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first ``n % world`` ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather ``[n_local, D]`` shards (possibly ragged) into ``[n_total, D]`` in rank order."""
    group = group if group is not None else work_group()
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


class PipelinedGather:
    """The all-gather of batch n overlapped with the forward of batch n + 1.

    ``push(local, n_total)`` starts a NON-BLOCKING ``all_gather_into_tensor`` of this rank's ``[n_local, ...]`` rows into one of
    two alternating buffers and hands back the finished matrix of the PREVIOUS push (``None`` the first time); ``flush()`` returns
    the last one.  On RCCL the collective runs on the communicator's own stream: it starts when the rows are ready and the
    compute stream only waits for it (``work.wait()``, a stream dependency, not a host block) one step later, so the 786 KB
    exchange and its launch latency hide under the next batch's kernels.  Rows keep rank order = clip order; ragged splits
    (``shard_bounds``) are padded to the widest shard and trimmed on the way out.  With one process nothing is exchanged and
    ``push`` returns the previous input.  The result tensors alias the two buffers: consume (or copy) one before the push after
    next overwrites it.
    """

    def __init__(self, group=None, force: bool = False, measure: bool = False) -> None:
        self.group = group if group is not None else work_group()
        # measure: time what the consumer actually WAITED for each gather (``exposed_ms``): on a GPU two events on the compute stream
        # around the stream dependency (0 when the collective had finished under the next batch's kernels), on the CPU the blocking wait
        self.measure = bool(measure)
        self._waits: List[Tuple[Any, Any]] = []
        self._cpu_wait_s = 0.0
        # force: go through the collective with a single rank too (scripts/rccl_one_rank.py: the only way to execute the RCCL leg on a
        # one-GPU box)
        self.active = dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self._bufs: List[Optional[torch.Tensor]] = [None, None]
        self._slot = 0
        self._pending: Optional[Tuple[Any, torch.Tensor, torch.Tensor, int, tuple]] = None      # (work, buffer, kept input, n_total, tail shape)

    def _buffer(self, slot: int, rows: int, width: int, like: torch.Tensor) -> torch.Tensor:
        b = self._bufs[slot]
        if b is None or b.shape != (rows, width) or b.dtype != like.dtype or b.device != like.device:
            b = torch.empty((rows, width), dtype=like.dtype, device=like.device)
            self._bufs[slot] = b
        return b

    def _finish(self) -> Optional[torch.Tensor]:
        if self._pending is None:
            return None
        work, buf, _kept, n_total, tail = self._pending
        self._pending = None
        if work is None:                       # single process: the "buffer" is the input itself
            return buf
        if self.measure and buf.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            self._waits.append((e0, e1))
        elif self.measure:
            import time
            t0 = time.perf_counter()
            work.wait()
            self._cpu_wait_s += time.perf_counter() - t0
        else:
            work.wait()
        sizes = [shard_bounds(n_total, r, self.world) for r in range(self.world)]
        width = max(hi - lo for lo, hi in sizes)
        if all(hi - lo == width for lo, hi in sizes):
            out = buf
        else:
            out = torch.cat([buf[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)
        return out.reshape((n_total,) + tail)

    def push(self, local: torch.Tensor, n_total: int) -> Optional[torch.Tensor]:
        done = self._finish()
        if not self.active:
            self._pending = (None, local, local, n_total, tuple(local.shape[1:]))
            return done
        tail = tuple(local.shape[1:])
        ncol = 1
        for d in tail:
            ncol *= int(d)
        flat = local.reshape(local.shape[0], ncol).contiguous()          # (an empty shard has no -1 to infer)
        sizes = [shard_bounds(n_total, r, self.world) for r in range(self.world)]
        width = max(hi - lo for lo, hi in sizes)
        if flat.shape[0] != width:             # ragged: pad to the widest shard
            padded = torch.zeros((width, flat.shape[1]), dtype=flat.dtype, device=flat.device)
            padded[: flat.shape[0]] = flat
            flat = padded
        buf = self._buffer(self._slot, self.world * width, flat.shape[1], flat)
        self._slot ^= 1
        work = dist.all_gather_into_tensor(buf, flat, group=self.group, async_op=True)
        self._pending = (work, buf, flat, n_total, tail)       # `flat` is kept alive until the collective has read it
        return done

    def flush(self) -> Optional[torch.Tensor]:
        return self._finish()

    def exposed_ms(self, reset: bool = True) -> float:
        """Milliseconds the consumer waited for gathers since the last reset (``measure=True``; synchronises the device)."""
        ms = 1e3 * self._cpu_wait_s
        if self._waits:
            torch.cuda.synchronize()
            ms += sum(a.elapsed_time(b) for a, b in self._waits)
        if reset:
            self._waits, self._cpu_wait_s = [], 0.0
        return ms


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
