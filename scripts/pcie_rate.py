#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: the caller hands over HOST buffers (fp32 wav, 640 KB per clip).
 (a) resident: wav already in HBM (what bench.py reports);
 (b) synchronous: pageable / pinned host tensor -> .to(device) -> forward, every step;
 (c) overlapped: avex_amd.extraction.extract_embeddings_in_memory with prefetch (pinned staging + side-stream copies)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
B, steps = 256, 6
cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", max_chunk_clips=256, residual="half")
host = 0.1 * torch.randn(B, 160000)
pinned = host.pin_memory()
dev = host.cuda()

def run(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)

print(f"(a) resident in HBM          : {run(lambda: enc.forward(dev, want_features=False, want_pooled=True)):8.0f} clips/s")
print(f"(b) pageable host, sync copy : {run(lambda: enc.forward(host.cuda(), want_features=False, want_pooled=True)):8.0f} clips/s")
print(f"(b) pinned host, sync copy   : {run(lambda: enc.forward(pinned.cuda(), want_features=False, want_pooled=True)):8.0f} clips/s")
# (c) double-buffered: copy of batch i+1 on a side stream while batch i computes
side = torch.cuda.Stream()
def overlapped():
    cur = pinned.cuda(non_blocking=True)
    outs = []
    for i in range(steps):
        with torch.cuda.stream(side):
            nxt = pinned.to("cuda", non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        outs.append(enc.forward(cur, want_features=False, want_pooled=True)["pooled"])
        torch.cuda.current_stream().wait_event(ev)
        cur = nxt
    return outs
overlapped(); torch.cuda.synchronize(); t0 = time.perf_counter(); overlapped(); torch.cuda.synchronize()
print(f"(c) pinned host, overlapped  : {B * steps / (time.perf_counter() - t0):8.0f} clips/s")
