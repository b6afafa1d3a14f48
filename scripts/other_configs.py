#!/usr/bin/env python3
"""The other BASELINE.json configs on one MI355X (parity-test cases, not bench lines): C1 single-clip latency of the BEATs path,
C3 EAT (512 x 5 s: frontend alone and the whole encoder), long BEATs clips, C4's per-GPU share is the bench itself, C5 see
scripts/effnet_bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth, kernels as K
from avex_amd.eat_audio_processor import EATAudioProcessor

def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

cfg = synth.BEATS_BASE_CFG
enc = K.BeatsEncoder(cfg, synth.beats_state_dict(cfg, seed=0), operand_dtype="f16", residual="half")
for B in (1, 4, 16):
    wav = (0.1 * torch.randn(B, 160000)).cuda()
    dt = timeit(lambda: enc.forward(wav, want_features=False, want_pooled=True), 20)
    print(f"C1 BEATs batch {B:2d} x 10 s: {1e3*dt:7.2f} ms per call, {B/dt:7.0f} clips/s")
proc = EATAudioProcessor()
wav = (0.1 * torch.randn(512, 80000)).cuda()
dt = timeit(lambda: proc(wav), 10)
print(f"C3 EAT frontend 512 x 5 s -> (512, 1024, 128): {1e3*dt:.2f} ms, {512/dt:.0f} clips/s, {512*(80000*4+1024*128*4)/dt/1e12:.2f} TB/s algorithmic")
from avex_amd.eat_encoder import EatEncoder
eat = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype="f16")
dt = timeit(lambda: eat.forward(wav, want_features=False, pooling="mean"), 5)
print(f"C3 EAT-base 512 x 5 s -> pooled 768-d: {1e3*dt:.1f} ms, {512/dt:.0f} clips/s, {512*97.0e9/dt/1e12:.0f} TFLOP/s")
del eat
for secs, B in ((20, 128), (60, 32)):
    wav = (0.1 * torch.randn(B, 16000 * secs)).cuda()
    dt = timeit(lambda: enc.forward(wav, want_features=False, want_pooled=True), 5)
    print(f"BEATs {B} clips x {secs} s ({enc.num_tokens(16000 * secs)} tokens each): {1e3*dt:.1f} ms, {B*secs/10/dt:.0f} clip-equivalents of 10 s per second")
