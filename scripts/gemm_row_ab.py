"""A/B of the attention output projection (K = 768 -> N = 768 at 126 976 rows, LayerNorm-folded residual, row statistics out) on the two
kernels: the 256-tile streaming kernel + ln_rowstats (variant 5) against the full-row kernel (variant 8, gemm_row.hip), alternating inside
one process, HIP-event timed; bit-identity asserted first.  Also fc2's shape (K = 3072) for the record.
    python scripts/gemm_row_ab.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avex_amd import kernels as K

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
E = 768


def case(M, Kd, name):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, Kd, generator=g).half().cuda()
    w = (torch.randn(E, Kd, generator=g) * Kd ** -0.5).half().cuda()
    x = torch.randn(M, E, generator=g).half().cuda()
    bias = torch.randn(E, generator=g).cuda(); gamma = (1 + 0.2 * torch.randn(E, generator=g)).cuda(); beta = (0.2 * torch.randn(E, generator=g)).cuda()
    xf = x.float(); rstd = 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)
    lrows = torch.zeros(M + (M & 1), 2, device="cuda"); lrows[:M, 0] = rstd; lrows[:M, 1] = -xf.mean(1) * rstd
    kw = dict(bias=bias, alpha=2.2, out_f32=False, out_half=True, rows_eps=1e-5, lnr_y=x, lnr_rows=lrows, lnr_gamma=gamma, lnr_beta=beta)
    r5 = K.gemm(a, w, variant=5, **kw); r8 = K.gemm(a, w, variant=8, **kw)
    same = torch.equal(r5["half"], r8["half"]) and torch.equal(r5["rows"][:M], r8["rows"][:M])
    print(f"{name}: M={M} K={Kd}: bit-identical {same}", flush=True)

    def timed(v, n=20):
        for _ in range(3):
            K.gemm(a, w, variant=v, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            K.gemm(a, w, variant=v, **kw)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for r in range(rounds):
        t5, t8 = timed(5), timed(8)
        fl = 2.0 * M * E * Kd
        print(f"  round {r}: variant 5 (+ ln_rowstats) {t5:7.1f} us ({fl / t5 / 1e6:6.0f} TF/s)   variant 8 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TF/s)   ratio {t8 / t5:.3f}", flush=True)


case(126976, 768, "out_proj")
case(126976, 3072, "fc2")
case(262656, 768, "EAT out_proj (512 x 513)")
