"""Where a tile of the full-row kernel (gemm_row.hip) spends its time: in-kernel s_memrealtime stamps of the diagnostic build
    AVEX_AMD_LIB_SUFFIX=rowst AVEX_AMD_EXTRA_CFLAGS=-DGEMM_ROW_STAMPS=1 python -m avex_amd.build
    AVEX_AMD_LIB=avex_amd/lib/libavexhip_rowst.so python scripts/gemm_row_stamps.py [K]
Per tile: prologue (first DMAs issued -> landed, barrier), K loop, epilogue; and the gap to the workgroup's next tile."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from avex_amd import kernels as K, _capi

Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 768
M, E = 126976, 768
g = torch.Generator().manual_seed(1)
a = torch.randn(M, Kd, generator=g).half().cuda(); w = (torch.randn(E, Kd, generator=g) * Kd ** -0.5).half().cuda()
x = torch.randn(M, E, generator=g).half().cuda(); bias = torch.randn(E, generator=g).cuda()
gamma = (1 + 0.2 * torch.randn(E, generator=g)).cuda(); beta = (0.2 * torch.randn(E, generator=g)).cuda()
lrows = torch.rand(M, 2, generator=g).cuda()
kw = dict(bias=bias, alpha=2.2, out_f32=False, out_half=True, rows_eps=1e-5, lnr_y=x, lnr_rows=lrows, lnr_gamma=gamma, lnr_beta=beta, variant=8)
for _ in range(50):
    K.gemm(a, w, **kw)
torch.cuda.synchronize()
L = _capi.lib()
nt = (M + 127) // 128
buf = np.zeros(4 * nt, np.uint64)
L.avexhip_debug_row_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.avexhip_debug_row_stamps(buf.ctypes.data, nt) == 0
st = buf.reshape(nt, 4).astype(np.float64) / 100.0      # us
t0 = st[:, 0].min()
pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
print(f"K={Kd}: {nt} tiles; kernel span {st[:, 3].max() - t0:.1f} us")
for name, v in (("prologue", pro), ("K loop", loop), ("epilogue", epi)):
    print(f"  {name:9s} median {np.median(v):6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}")
for rnd in range(4):
    sel = slice(256 * rnd, min(256 * (rnd + 1), nt))
    print(f"  round {rnd}: starts {np.median(st[sel, 0]) - t0:6.1f}  loop {np.median(loop[sel]):6.2f}  epilogue {np.median(epi[sel]):6.2f}  ends {np.median(st[sel, 3]) - t0:6.1f} us")
