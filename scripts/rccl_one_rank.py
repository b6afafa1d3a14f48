#!/usr/bin/env python3
"""The RCCL leg of the multi-GPU loop on a ONE-GPU box: backend "nccl" with a single rank, the bench's own step (forward + non-blocking
all-gather through avex_amd.dist.PipelinedGather, forced on) -- what a one-GPU box can execute of SURVEY.md section 8e: RCCL initialises,
the asynchronous collective on the communicator's stream is ordered against this library's kernels on torch's current stream, and every
gathered matrix is the forward it belongs to, bit for bit.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_one_rank.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from avex_amd import synth, kernels as K
from avex_amd.dist import PipelinedGather

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
dist.init_process_group(backend="nccl", device_id=dev)
print(f"backend {dist.get_backend()}, world {dist.get_world_size()}, RCCL/NCCL version {torch.cuda.nccl.version()}", flush=True)
cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
B = 256
wavs = [torch.from_numpy(synth.noise_clips(B, 160000, seed=0, first_clip=1000 * i)).to(dev) for i in range(3)]
want = [enc.forward(w, want_features=False, want_pooled=True)["pooled"].clone() for w in wavs]
pipe = PipelinedGather(force=True)
assert pipe.active and pipe.world == 1
steps, bad = 12, 0
dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    got = pipe.push(enc.forward(wavs[i % 3], want_features=False, want_pooled=True)["pooled"], B)
    if got is not None:
        bad += int(not torch.equal(got, want[(i - 1) % 3]))
got = pipe.flush()
bad += int(not torch.equal(got, want[(steps - 1) % 3]))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
t = torch.tensor([dt], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print(f"{steps} steps of forward + non-blocking all-gather over RCCL: {1e3 * dt:.2f} ms per step ({B / dt:.0f} clips/s), "
      f"{bad} gathered matrices differed from their forward")
g = torch.empty((B, 768), dtype=torch.float32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dist.all_gather_into_tensor(g, want[0]); torch.cuda.synchronize()
e0.record()
for _ in range(10):
    dist.all_gather_into_tensor(g, want[0])
e1.record(); torch.cuda.synchronize()
print(f"blocking all_gather_into_tensor of [{B}, 768] fp32 with one rank: {e0.elapsed_time(e1) / 10:.3f} ms; equal to its input: {torch.equal(g, want[0])}")
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
