"""Time the device probe heads behind the encoder at the bench batch (256 clips x 10 s): linear probe on mean-pooled taps, MLP
probe, attention probe on the full [B, 496, 768] taps.  Prints ms per batch for the head alone (embeddings resident)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from avex_amd import kernels as K
from avex_amd import probes as P


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B, T, D, C, L = 256, 496, 768, 50, 4
    g = torch.Generator().manual_seed(0)
    pooled = [torch.randn(B, D, generator=g).cuda() for _ in range(L)]
    taps = [torch.randn(B, T, D, generator=g).cuda() for _ in range(L)]
    lin = P.LinearProbe(None, [], C, feature_mode=True, input_dim=[(D,)] * L)
    mlp = P.MLPProbe(None, [], C, feature_mode=True, input_dim=[(D,)] * L)
    att = P.AttentionProbe(None, [], C, feature_mode=True, input_dim=[(T, D)] * L, aggregation="none", num_heads=8, num_layers=2)
    print(f"linear probe  ({L} pooled taps -> {C} classes): {timeit(lambda: lin(pooled)):.3f} ms / batch of {B}")
    print(f"mlp probe     (512, 256 hidden):               {timeit(lambda: mlp(pooled)):.3f} ms")
    print(f"layer mix     ({L} x [B,{T},{D}] fp32):          {timeit(lambda: K.layer_mix(taps)):.3f} ms   ({(L + 1) * B * T * D * 4 / 1e9:.2f} GB)")
    x = taps[0]
    w = torch.randn(3 * D, D, generator=g).cuda() * D ** -0.5
    t = timeit(lambda: K.dense_f32(x, w))
    print(f"dense_f32     [{B * T}, {D}] x [{3 * D}, {D}]^T:     {t:.3f} ms   ({2 * B * T * D * 3 * D / t / 1e9:.1f} TFLOP/s fp32)")
    qkv = K.dense_f32(x, w)
    t = timeit(lambda: K.mha_f32(qkv, 8), 3)
    print(f"mha_f32       8 heads of 96 over {T} tokens:      {t:.3f} ms   ({4 * B * 8 * T * T * 96 / t / 1e9:.1f} TFLOP/s fp32)")
    print(f"attention probe (2 layers, all of the above):    {timeit(lambda: att(taps), 3):.3f} ms")


if __name__ == "__main__":
    main()
