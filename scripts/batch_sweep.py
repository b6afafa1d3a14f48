#!/usr/bin/env python3
"""BEATs-base throughput against the batch handed to one forward: does a working set that fits the 256 MB Infinity Cache (fewer clips per call)
buy more than the GEMMs' tile-count rounding costs?  Steady state: each batch size loops for ~3 s.    python scripts/batch_sweep.py [sizes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K

sizes = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 96, 128, 160, 192, 256, 384, 512]
cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
wav_all = (0.1 * torch.randn(max(sizes), 160000)).cuda()
for rep in range(2):
    for B in sizes:
        wav = wav_all[:B].contiguous()
        f = lambda: enc.forward(wav, want_features=False, want_pooled=True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        n = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.5:
            for _ in range(5): f()
            n += 5
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"batch {B:4d}: {1e3*dt:8.3f} ms per forward, {B/dt:8.0f} clips/s", flush=True)
