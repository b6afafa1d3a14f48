#!/usr/bin/env python3
"""The EfficientNet mel frontend alone (800-point STFT by FFT, 128 mels, log, min-max normalisation): ms per 256 clips of 10 s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avex_amd import kernels as K
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
wav = (0.1 * torch.randn(256, 160000)).cuda()
for _ in range(3): plan(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): plan(wav)
torch.cuda.synchronize(); print(f"mel frontend, 256 clips x 10 s: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
