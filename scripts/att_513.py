#!/usr/bin/env python3
"""EAT's 513 tokens (512 patches + the class token) against 512 and 496: the attention kernel alone at 256 clips x 12 heads, no bias table.
513 runs the long-clip instantiation (keys in blocks of 256: a third key block with ONE key) plus the one-wave-per-row tail kernel for the
513th query row."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K

B, H = 256, 12
E = 64 * H


def run(T, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    qkv = (torch.randn(B * T, 3 * E, device="cuda") * 1.0).half()
    for _ in range(3):
        K.attention(qkv, B, T, H, None, None, None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        K.attention(qkv, B, T, H, None, None, None, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    for k in (env or {}):
        os.environ.pop(k, None)
    fl = 4.0 * B * H * T * T * 64
    print(f"T = {T:4d} {str(env or ''):40s}: {ms:7.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s")


run(496)
run(512)
run(513)
run(513, {"AVEX_AMD_ATT_NO_TAIL": "1"})
run(513, {"AVEX_AMD_ATT_NO_XT": "1"})
run(520)
run(520, {"AVEX_AMD_ATT_NO_TAIL": "1"})
run(544)
run(544, {"AVEX_AMD_ATT_NO_TAIL": "1"})
run(768)
run(1024)
