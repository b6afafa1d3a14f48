import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from avex_amd import kernels as K
B, H = 256, 12
E = 64 * H
def run(T, scale=1.0, zero=False):
    qkv = (torch.randn(B * T, 3 * E, device="cuda") * scale).half()
    if zero: qkv.zero_()
    for _ in range(3): K.attention(qkv, B, T, H, None, None, None, None)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
    for _ in range(n): K.attention(qkv, B, T, H, None, None, None, None)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"T={T} scale={scale} zero={zero}: {ms:.3f} ms  {4.0*B*H*T*T*64/ms/1e9:.0f} TF/s")
for T in (384, 448, 480, 488, 496, 504, 512):
    run(T)
run(496, 0.3); run(512, 0.3); run(496, zero=True); run(512, zero=True)
