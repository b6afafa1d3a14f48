#!/usr/bin/env python3
"""AVES (wav2vec2-base, 12 layers) throughput on one MI355X: clips/s for 10 s clips, synthetic weights (not a BASELINE config)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth
from avex_amd.aves_encoder import AvesEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = AvesEncoder(synth.AVES_BASE_CFG, synth.aves_state_dict())
wav = (0.1 * torch.randn(B, 160000)).cuda()
for _ in range(2): enc.forward(wav, want_features=False, want_pooled=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 4
for _ in range(n): out = enc.forward(wav, want_features=False, want_pooled=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
enc.extract_conv_features(wav)      # (builds its packed weights on the first call)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(n): enc.extract_conv_features(wav)
torch.cuda.synchronize(); dc = (time.perf_counter() - t1) / n
gflop = 2 * (31999 * 512 * 10 + (15999 + 7999 + 3999 + 1999) * 512 * 1536 + (999 + 499) * 512 * 1024) / 1e9 + 85.0
print(f"AVES B={B}: {1e3*dt:.1f} ms/step, {B/dt:.0f} clips/s (conv feature extractor {1e3*dc:.1f} ms); ~{gflop:.0f} GFLOP/clip -> {B/dt*gflop/1e3:.0f} TFLOP/s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
enc.set_profiling(True)
enc.forward(wav, want_features=False, want_pooled=True)
torch.cuda.synchronize()
print("stages (ms, serialised HIP events): " + "  ".join(f"{n} {ms:.3f}" for n, ms, _ in enc.last_profile()))
