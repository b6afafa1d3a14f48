#!/usr/bin/env python3
"""EAT-base on one MI355X at BASELINE config C3's shape: 512 clips x 5 s @ 16 kHz -> [512, 513, 768] -> pooled 768-d."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import synth
from avex_amd.eat_encoder import EatEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
enc = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype=os.environ.get("AVEX_AMD_OPERAND", "f16"))
wav = torch.from_numpy(synth.noise_clips(B, 80000, seed=0)).cuda()
def step():
    return enc.forward(wav, want_features=False, pooling="mean")["pooled"]
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
for _ in range(n): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
# per clip: patch GEMM 2*512*256*768 + 12 * (2*513*768*(2304+768+3072+3072) + 4*513*513*768)
fl = 2 * 512 * 256 * 768 + 12 * (2 * 513 * 768 * (2304 + 768 + 3072 + 3072) + 4 * 513 * 513 * 768)
print(f"EAT-base B={B} x 5 s: {1e3*dt:.1f} ms/step, {B/dt:.0f} clips/s, {B*fl/dt/1e12:.0f} TFLOP/s ({fl/1e9:.1f} GFLOP/clip); "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB; out {tuple(out.shape)}")
enc.set_profiling(True)
step()
torch.cuda.synchronize()
print("stages (ms, serialised HIP events): " + "  ".join(f"{n} {ms:.3f}" for n, ms, _ in enc.last_profile()))
enc.set_profiling(False)
