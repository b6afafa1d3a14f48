import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import kernels as K, _capi
M, N, Kd = 256 * 496, 2304, int(sys.argv[1]) if len(sys.argv) > 1 else 768
x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); bias = torch.randn(N, device="cuda")
L = _capi.lib()
for _ in range(3): K.gemm(x, w, bias=bias, out_f32=False, out_half=True, variant=2)
torch.cuda.synchronize()
L.avexhip_debug_gemm_stamps(1, None, 0)
K.gemm(x, w, bias=bias, out_f32=False, out_half=True, variant=2)
torch.cuda.synchronize()
nb = (M // 256) * (N // 256)
buf = np.zeros(4 * nb, np.uint64)
L.avexhip_debug_gemm_stamps(0, buf.ctypes.data, nb)
t = buf.reshape(nb, 4).astype(np.float64) / 100.0   # us
t0 = t[:, 0].min()
d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]], 1)
print(f"K={Kd}: blocks {nb}; kernel span {t[:,3].max() - t0:.1f} us")
print("per-block us (median / p10 / p90): prologue %.2f/%.2f/%.2f  loop %.2f/%.2f/%.2f  epilogue %.2f/%.2f/%.2f  total %.2f/%.2f/%.2f" % tuple(
    v for col in range(4) for v in (np.median(d[:, col]), np.percentile(d[:, col], 10), np.percentile(d[:, col], 90))))
# gaps between consecutive blocks on the same CU are not visible here; estimate from span: span / rounds vs median total
print(f"rounds {nb/256:.2f}; span/rounds {(t[:,3].max() - t0)/(nb/256):.2f} us vs median block total {np.median(d[:,3]):.2f} us")
