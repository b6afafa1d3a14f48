#!/usr/bin/env python3
"""What hook taps cost at the bench shape (256 x 10 s): pooled output only; + the last layer's tap; + all 13 taps, mean-pooled on the
device (extract_embeddings(aggregation="mean") with layers "all") and unpooled (aggregation="none": 13 x 390 MB of fp32 taps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K

cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wav = (0.1 * torch.randn(B, 160000)).cuda()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


cases = [("pooled features only", dict(want_features=False, want_pooled=True)),
         ("+ last-layer tap, mean-pooled", dict(hook_layers=[12], hook_pooled=True, want_features=False, want_pooled=True)),
         ("+ all 13 taps, mean-pooled", dict(hook_layers=list(range(13)), hook_pooled=True, want_features=False, want_pooled=True)),
         ("+ all 13 taps, max over tokens", dict(hook_layers=list(range(13)), hook_pooled="max", want_features=False, want_pooled=True)),
         ("+ all 13 taps, first token", dict(hook_layers=list(range(13)), hook_pooled="cls_token", want_features=False, want_pooled=True)),
         ("+ last-layer tap, unpooled", dict(hook_layers=[12], want_features=False, want_pooled=True)),
         ("+ all 13 taps, unpooled", dict(hook_layers=list(range(13)), want_features=False, want_pooled=True)),
         ("features [B, 496, 768] fp32", dict(want_features=True, want_pooled=False))]
base = None
for name, kw in cases:
    ms = timeit(lambda: enc.forward(wav, **kw))
    base = base or ms
    print(f"{name:36s} {ms:8.2f} ms per {B} clips  ({B / ms * 1e3:7.0f} clips/s, {ms / base:5.3f}x)")
enc.set_profiling(True)
enc.forward(wav, hook_layers=list(range(13)), hook_pooled=True, want_features=False, want_pooled=True)
print("stages, all taps mean-pooled: " + "  ".join(f"{n} {t:.3f}" for n, t, _ in enc.last_profile()))
