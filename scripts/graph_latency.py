#!/usr/bin/env python3
"""Small-batch latency with the forward captured in a HIP graph (torch.cuda.CUDAGraph stream capture of the C-ABI launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
for B in (1, 4, 16):
    wav = (0.1 * torch.randn(B, 160000)).cuda()
    for _ in range(3): ref = enc.forward(wav, want_features=False, want_pooled=True)["pooled"].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        enc.forward(wav, want_features=False, want_pooled=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(out, ref)
    t0 = time.perf_counter(); n = 50
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): enc.forward(wav, want_features=False, want_pooled=True)
    torch.cuda.synchronize(); de = (time.perf_counter() - t0) / n
    print(f"batch {B:2d}: eager {1e3*de:.2f} ms, graph replay {1e3*dg:.2f} ms, identical output: {ok}")
