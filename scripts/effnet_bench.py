#!/usr/bin/env python3
"""EfficientNet-B0 on mel spectrograms (BASELINE config C5 shape: 10 s clips, n_fft 800 / hop 160 / 128 mels) on one MI355X."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
from avex_amd.effnet_encoder import EfficientNetB0Encoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc = EfficientNetB0Encoder(synth.effnet_b0_state_dict())
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
wav = (0.1 * torch.randn(B, 160000)).cuda()
def step():
    return enc.forward(plan(wav), want_features=False, want_pooled=True)["pooled"]
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 4
for _ in range(n): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"EfficientNet-B0 B={B}: {1e3*dt:.1f} ms/step, {B/dt:.0f} clips/s (wav -> mel -> features -> pooled 1280-d); peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB; out {tuple(out.shape)}")
