#!/usr/bin/env python3
"""Two ranks over RCCL on a ONE-GPU box (both on cuda:0), if RCCL allows it: rendezvous, communicator set-up between two processes (dmabuf IPC),
avex_amd.dist.init_distributed with its working group, and the bench's non-blocking all-gather with different rows per rank.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 scripts/rccl_two_ranks_one_gpu.py
RCCL normally refuses two ranks on one device ("Duplicate GPU detected"); the script reports that and exits 0 -- it is a probe, not a test."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NCCL_DEBUG", "WARN")
os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
os.environ["LOCAL_RANK"] = "0"                      # both ranks on device 0
import torch
import torch.distributed as dist
from avex_amd import dist as D

rank = int(os.environ.get("RANK", "0"))
try:
    r, world, lr = D.init_distributed("nccl")
    print(f"[rank {rank}] init_distributed ok: world {world}, local device {lr}, work group {D.work_group() is not None}", flush=True)
    dev = torch.device("cuda", 0)
    B, Dm = 256, 768
    pipe = D.PipelinedGather(force=True)
    want = [torch.full((B, Dm), float(100 * s + rank), device=dev) for s in range(4)]
    bad = 0
    for s in range(4):
        got = pipe.push(want[s], B)
        if got is not None:
            exp = torch.cat([torch.full((B, Dm), float(100 * (s - 1) + q), device=dev) for q in range(world)])
            bad += int(not torch.equal(got, exp))
    got = pipe.flush()
    exp = torch.cat([torch.full((B, Dm), float(300 + q), device=dev) for q in range(world)])
    bad += int(not torch.equal(got, exp))
    rows = D.all_gather_rows(torch.full((3 + rank, 4), float(rank), device=dev), 3 * world + sum(range(world)))
    print(f"[rank {rank}] 4 pipelined gathers + a ragged gather over RCCL with {world} ranks on one GPU: {bad} wrong; ragged rows {tuple(rows.shape)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"[rank {rank}] RCCL with two ranks on one device did not come up: {type(e).__name__}: {str(e)[:300]}", flush=True)
