#!/usr/bin/env python3
"""Throughput of the EfficientNet mel frontend (n_fft 800, hop 160, 128 mels, log + min-max) on one MI355X."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
wav = (0.1 * torch.randn(B, 160000)).cuda()
for _ in range(2): plan(wav)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
for _ in range(n): y = plan(wav)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
gb = B * (160000 * 4 + 128 * 1001 * 4 * 3) / 1e9      # read wav, write + read + write the log-mel (normalisation pass)
fl = B * 1001 * 800 * 896 * 2 / 1e12
print(f"melspec B={B}: {1e3*dt:.2f} ms, {B/dt:.0f} clips/s, {gb/dt/1e3:.2f} TB/s algorithmic HBM, dense DFT {fl/dt:.1f} TFLOP/s fp32 (MFMA f32 peak 157)")
