#!/usr/bin/env python3
"""The encoder's four products at a few clips (1 ... 64 x 496 rows): the 128-tile kernel (with split-K where K is long) against the 256-tile
streaming kernel, kernel time in microseconds.  Each timed launch sits between two events that were queued behind a long blocker product,
so the host's launch cost is not in the figures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from avex_amd import kernels as K

torch.manual_seed(0)
dev = "cuda"
big_a = torch.randn(65536, 3072, device=dev).half()
big_w = torch.randn(3072, 3072, device=dev).half() * 0.02
big_b = torch.zeros(3072, device=dev)


def blocker():
    for _ in range(3):
        K.gemm(big_a, big_w, bias=big_b, out_f32=False, out_half=True)


def timed(fn, n=12):
    fn(); fn()
    torch.cuda.synchronize()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    blocker()
    for i in range(n):
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in zip(s, e))
    return t[len(t) // 2] * 1e3


shapes = [("qkv", 768, 2304, False), ("out_proj", 768, 768, True), ("fc1", 768, 3072, False), ("fc2", 3072, 768, True)]
print(f"{'rows':>6s} " + " ".join(f"{n + ' 128':>13s} {n + ' 256':>13s}" for n, *_ in shapes))
for clips in (1, 2, 4, 8, 16, 32, 64):
    M = clips * 496
    row = [f"{M:6d}"]
    for name, Kd, N, resid in shapes:
        a = torch.randn(M, Kd, device=dev).half()
        w = (torch.randn(N, Kd, device=dev) * 0.03).half()
        b = torch.zeros(N, device=dev)
        r = torch.randn(M, N, device=dev).half() if resid else None
        kw = dict(bias=b, out_f32=False, out_half=True, resid_half=r, gelu=(name == "fc1"))
        t3 = timed(lambda: K.gemm(a, w, variant=3, splitk=(Kd >= 1024), **kw))
        t5 = timed(lambda: K.gemm(a, w, variant=5, **kw))
        row.append(f"{t3:13.1f} {t5:13.1f}")
    print(" ".join(row))
