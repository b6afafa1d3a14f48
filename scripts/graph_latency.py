#!/usr/bin/env python3
"""Small batches are launch-bound: the eager forward (one C call, ~95 kernel launches) against the same forward replayed from a
hipGraph (BeatsEncoder.capture), milliseconds per call, 10 s clips, pooled output only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
print("batch | eager ms/call  clips/s | graph ms/call  clips/s | nodes | speed-up | latency of ONE synchronous call: eager, graph (ms)")
for B in (1, 2, 4, 8, 16, 32, 64):
    wav = (0.1 * torch.randn(B, 160000)).cuda()
    g = enc.capture(B, 160000, want_features=False, want_pooled=True)
    g.wav.copy_(wav)
    te = timeit(lambda: enc.forward(wav, want_features=False, want_pooled=True), 50)
    tg = timeit(lambda: g.replay(), 50)

    def sync_call(fn):
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    le = sync_call(lambda: enc.forward(wav, want_features=False, want_pooled=True))
    lg = sync_call(lambda: g.replay())

    def host_cost(fn):      # time the submitting thread spends inside the call (device idle before, not waited for after)
        ts = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        return sorted(ts)[len(ts) // 2]
    he = host_cost(lambda: enc.forward(wav, want_features=False, want_pooled=True))
    hg = host_cost(lambda: g.replay())
    print(f"{B:5d} | {1e3*te:8.3f} {B/te:9.0f} | {1e3*tg:8.3f} {B/tg:9.0f} | {g.nodes:5d} | {te/tg:5.2f}x | {1e3*le:.3f} {1e3*lg:.3f} | host thread inside the call: {1e3*he:.3f} {1e3*hg:.3f} ms")
    g.close()
