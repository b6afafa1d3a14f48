#!/usr/bin/env python3
"""In-kernel stamps of the persistent 256-tile GEMM (variant 5): per tile wait / K loop / epilogue times and the
shader clock held inside the K loop.   python scripts/gemm_stamps5.py [K] [N] [gelu|half|resid]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import kernels as K, _capi
Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
mode = sys.argv[3] if len(sys.argv) > 3 else "half"
M = 256 * 496
x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); bias = torch.randn(N, device="cuda")
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw = dict(bias=bias, out_f32=False, out_half=True, variant=variant)
if mode == "gelu": kw["gelu"] = True
if mode == "resid": kw["resid_half"] = torch.randn(M, N, device="cuda").half(); kw["alpha"] = 2.2
L = _capi.lib()
for _ in range(200): K.gemm(x, w, **kw)      # let the clock settle under load
torch.cuda.synchronize()
L.avexhip_debug_gemm_stamps(1, None, 0)
K.gemm(x, w, **kw)
torch.cuda.synchronize()
nt = min((M // 256) * (N // 256), 8192)
buf = np.zeros(4 * nt, np.uint64); clk = np.zeros(2 * nt, np.uint64)
L.avexhip_debug_gemm_stamps(0, buf.ctypes.data, nt)
L.avexhip_debug_gemm_clocks(clk.ctypes.data, nt)
t = buf.reshape(nt, 4).astype(np.float64) / 100.0
c = clk.reshape(nt, 2).astype(np.float64)
d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
ghz = (c[:, 1] - c[:, 0]) / (t[:, 2] - t[:, 1]) / 1e3
print(f"variant {variant} K={Kd} N={N} {mode}: tiles {nt}; span {t[:,3].max() - t[:,0].min():.1f} us")
for name, col in (("wait", 0), ("loop", 1), ("epilogue", 2), ("tile", 3)):
    print(f"  {name:9s} median {np.median(d[:, col]):6.2f}  p10 {np.percentile(d[:, col], 10):6.2f}  p90 {np.percentile(d[:, col], 90):6.2f} us")
print(f"  loop clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f}); loop cycles/K-tile {np.median(c[:,1]-c[:,0])/(Kd//64):.0f}")
kc = np.zeros(256 * 64, np.uint64)
L.avexhip_debug_gemm_kclocks(kc.ctypes.data, 256)
kc = kc.reshape(256, 64).astype(np.float64)
nk = Kd // 64
if nk < 64 and variant == 5:
    dk = np.diff(kc[:, : nk + 1], axis=1)
    print("  cycles per K-tile by position in the tile (median over 256 workgroups, third tile):")
    print("   ", " ".join(f"{v:.0f}" for v in np.median(dk, axis=0)))
