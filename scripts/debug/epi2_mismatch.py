"""Where do the fast residual epilogue (EPI 2, with and without row statistics) and the generic one disagree?  Rows / columns of the mismatches."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from avex_amd import kernels as K
torch.manual_seed(0)
M, N, Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 2500, 768, 256
a = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); bias = torch.randn(N, device="cuda")
rh = torch.randn(M, N, device="cuda").half()
def run(env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return K.gemm(a, w, bias=bias, variant=5, out_f32=False, out_half=True, resid_half=rh, alpha=2.2133638, **kw)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
for stats in (False, True):
    ref = run({"AVEX_AMD_GEMM_GENERIC": "1"}, stats_out=stats)
    for name, env in (("fast", {}), ("fast grid 8", {"AVEX_AMD_GEMM_GRID": "8"}), ("fast grid 8 again", {"AVEX_AMD_GEMM_GRID": "8"}), ("generic grid 8", {"AVEX_AMD_GEMM_GENERIC": "1", "AVEX_AMD_GEMM_GRID": "8"})):
        r = run(env, stats_out=stats)
        for key in ref:
            d = (r[key] != ref[key])
            if key == "stats": d = d.any(-1)
            n = int(d.sum())
            msg = f"stats={stats} {name:18s} {key:5s} mismatches {n}"
            if n:
                rows = d.any(1).nonzero().flatten().cpu().numpy(); cols = d.any(0).nonzero().flatten().cpu().numpy()
                msg += f"  rows {rows[:12]}..{rows[-3:]} ({len(rows)})  cols {cols[:12]}..{cols[-3:]} ({len(cols)})"
                i, j = int(rows[0]), int(cols[0])
                msg += f"  e.g. [{i},{j}] got {r[key][i, j].flatten()[:2].tolist()} want {ref[key][i, j].flatten()[:2].tolist()}"
            print(msg, flush=True)
