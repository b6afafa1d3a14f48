#!/usr/bin/env python3
"""The four layer GEMMs in the forms the 256-clip step launches them (LayerNorm fold on): time per launch, and with the diagnostic
library (AVEX_AMD_LIB=avex_amd/lib/libavexhip_diag.so) the in-kernel stamps: K loop / epilogue per tile, cycles per K-tile by position.

    python scripts/gemm_forms.py [--shapes qkv,out,fc1,fc2] [--iters 30] [--variant 0] [--stamps]

qkv  K 768 -> N 2304, A rows raw + folded LayerNorm (EPI 1, LN)            backbone.py:531-533
out  K 768 -> N 768,  residual = LayerNorm(y) on the fly + row statistics  backbone.py:572
fc1  K 768 -> N 3072, folded LayerNorm + GELU                              backbone.py:365-368
fc2  K 3072 -> N 768, residual fold + row statistics                       backbone.py:370
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avex_amd import kernels as K, _capi

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="qkv,out,fc1,fc2")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--plain", action="store_true", help="no LayerNorm fold (the forms of scripts/gemm_bench.py)")
ap.add_argument("--rounds", type=int, default=1, help="repeat the whole table (interleaved A/B inside one process: set env knobs per round outside)")
a = ap.parse_args()
M = a.clips * 496
dev = "cuda"
torch.manual_seed(0)


def make(name):
    N, Kd = {"qkv": (2304, 768), "out": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}[name]
    x = torch.randn(M, Kd, device=dev).half()
    w = (torch.randn(N, Kd, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias, out_f32=False, out_half=True, variant=a.variant)
    rows = torch.stack([torch.rand(M + 1, device=dev) + 0.5, torch.randn(M + 1, device=dev) * 0.1], 1).contiguous()
    if name in ("qkv", "fc1"):
        if not a.plain:
            kw.update(ln_rows=rows, ln_s=torch.randn(N, device=dev))
        if name == "fc1":
            kw["gelu"] = True
    else:
        kw["alpha"] = 2.2
        if a.plain:
            kw["resid_half"] = torch.randn(M, N, device=dev).half()
        else:
            kw.update(lnr_y=torch.randn(M, N, device=dev).half(), lnr_rows=rows[:M].contiguous(), lnr_gamma=torch.rand(N, device=dev) + 0.5,
                      lnr_beta=torch.randn(N, device=dev), stats_out=True)
    return x, w, kw, N, Kd


L = _capi.lib()
for rnd in range(a.rounds):
    for name in a.shapes.split(","):
        x, w, kw, N, Kd = make(name)
        for _ in range(5):
            K.gemm(x, w, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            K.gemm(x, w, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tiles = (M // 256) * (N // 256)
        print(f"{name:4s} M={M} N={N} K={Kd}: {ms*1e3:8.1f} us  {2.0*M*N*Kd/ms/1e9:7.1f} TFLOP/s   ({tiles/256:.2f} tiles per CU, {ms*1e3/np.ceil(tiles/256):.1f} us per round)", flush=True)
        if a.stamps:
            for _ in range(100):
                K.gemm(x, w, **kw)
            torch.cuda.synchronize()
            L.avexhip_debug_gemm_stamps(1, None, 0)
            K.gemm(x, w, **kw)
            torch.cuda.synchronize()
            nt = min(tiles, 8192)
            buf = np.zeros(4 * nt, np.uint64); clk = np.zeros(2 * nt, np.uint64)
            L.avexhip_debug_gemm_stamps(0, buf.ctypes.data, nt)
            L.avexhip_debug_gemm_clocks(clk.ctypes.data, nt)
            t = buf.reshape(nt, 4).astype(np.float64) / 100.0
            c = clk.reshape(nt, 2).astype(np.float64)
            ok = t[:, 3] > 0
            t, c = t[ok], c[ok]
            d = np.stack([t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
            ghz = (c[:, 1] - c[:, 0]) / (t[:, 2] - t[:, 1]) / 1e3
            for nm, col in (("loop", 0), ("epilogue", 1), ("tile", 2)):
                print(f"     {nm:9s} median {np.median(d[:, col]):6.2f}  p10 {np.percentile(d[:, col], 10):6.2f}  p90 {np.percentile(d[:, col], 90):6.2f} us")
            print(f"     loop clock {np.median(ghz):.3f} GHz; loop cycles per K-tile {np.median(c[:,1]-c[:,0])/(Kd//64):.0f} (MFMA floor 2048); span {t[:,3].max()-t[:,0].min():.1f} us")
            nk = Kd // 64
            if nk < 63:
                kc = np.zeros(256 * 64, np.uint64)
                L.avexhip_debug_gemm_kclocks(kc.ctypes.data, 256)
                dk = np.diff(kc.reshape(256, 64).astype(np.float64)[:, : nk + 1], axis=1)
                print("     cycles per K-tile by position (third tile, median over workgroups):", " ".join(f"{v:.0f}" for v in np.median(dk, axis=0)))
