#!/usr/bin/env python3
"""Randomised checks of the GEMM (every kernel variant and epilogue the model uses, ragged M, forced small grids) and of the attention
kernels (random token counts on both sides of 512, heads, key padding, bias on / off, gate on / off; the plain attention for heads of
32 / 64 / 96 / 128) against fp64 / NumPy.
    python tests/tools/fuzz_kernels.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from avex_amd import kernels as K
from oracle import beats_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def rnd_half(x, dt):
    t = torch.from_numpy(x).to(torch.float16 if dt == "f16" else torch.bfloat16)
    return t.float().numpy(), t

worst = {"gemm": 0.0, "attention": 0.0, "attention_hd": 0.0}
for c in range(cases):
    dt = ["f16", "bf16"][int(rng.integers(0, 4) == 0)]
    # ---------------- GEMM ----------------
    M = int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000), rng.integers(3000, 9000)]))
    N = int(rng.choice([128, 256, 384, 512, 768, 1024, 2304, 3072]))
    Kd = int(rng.choice([64, 128, 192, 256, 512, 768, 3072]))
    variant = int(rng.choice([0, 0, 2, 3, 5]))
    if rng.integers(0, 3) == 0:
        os.environ["AVEX_AMD_GEMM_GRID"] = str(int(rng.choice([8, 16, 64])))
    else:
        os.environ.pop("AVEX_AMD_GEMM_GRID", None)
    a, ad = rnd_half(rng.standard_normal((M, Kd)).astype(np.float32), dt)
    w, wd = rnd_half((0.05 * rng.standard_normal((N, Kd))).astype(np.float32), dt)
    bias = (0.1 * rng.standard_normal(N)).astype(np.float32)
    rh, rhd = rnd_half(rng.standard_normal((M, N)).astype(np.float32), dt)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    ad, wd, rhd, bd = ad.cuda(), wd.cuda(), rhd.cuda(), torch.from_numpy(bias).cuda()
    mode = int(rng.integers(0, 6))
    tolh = 7e-4 if dt == "f16" else 5e-3
    fold_ok = N % 256 == 0 and Kd >= 128 and M >= 2
    if mode >= 4 and not fold_ok:
        mode -= 2
    if mode >= 4:
        # folded LayerNorm (round 3): the producer writes y = rh * alpha + a @ w.T + bias with row statistics; mode 4 consumes LN(y) as
        # the A operand of a second product through folded weights, mode 5 as the residual of one; against the two-pass fp64 LayerNorm of
        # the stored rows.  Every second case through the generic epilogue.
        generic = bool(rng.integers(0, 2))
        alpha = 2.2133638
        p = K.gemm(ad, wd, bias=bd, resid_half=rhd, alpha=alpha, out_f32=False, out_half=True, stats_out=True, variant=5)
        y = p["half"].float().cpu().numpy().astype(np.float64)
        rows = K.ln_rowstats(p["stats"])
        gamma = (1.0 + 0.2 * rng.standard_normal(N)).astype(np.float32); beta = (0.2 * rng.standard_normal(N)).astype(np.float32)
        mu = y.mean(1, keepdims=True); rstd = 1.0 / np.sqrt(y.var(1, keepdims=True) + 1e-5)
        ln = (y - mu) * rstd * gamma + beta
        N2 = int(rng.choice([256, 512, 768]))
        if generic:
            os.environ["AVEX_AMD_GEMM_GENERIC"] = "1"
        if mode == 4:
            w2 = (0.05 * rng.standard_normal((N2, N))).astype(np.float32); b2 = (0.1 * rng.standard_normal(N2)).astype(np.float32)
            w2f, w2fd = rnd_half((w2 * gamma[None, :]).astype(np.float32), dt)
            s2 = w2f.astype(np.float64).sum(1).astype(np.float32)
            b2f = (b2 + w2.astype(np.float64) @ beta).astype(np.float32)
            act = bool(rng.integers(0, 2))
            r = K.gemm(p["half"], w2fd.cuda(), bias=torch.from_numpy(b2f).cuda(), gelu=act, out_f32=False, out_half=True, ln_rows=rows, ln_s=torch.from_numpy(s2).cuda())
            want = ((y - mu) * rstd) @ w2f.astype(np.float64).T + b2f
            if act:
                want = O.gelu_erf(want.astype(np.float32))
            e = rel(r["half"].float().cpu().numpy(), want) / (1.0e-3 if dt == "f16" else 6e-3)
        else:
            a2, a2d = rnd_half(rng.standard_normal((M, 128)).astype(np.float32), dt)
            w2, w2d = rnd_half((0.05 * rng.standard_normal((N, 128))).astype(np.float32), dt)
            b2 = (0.1 * rng.standard_normal(N)).astype(np.float32)
            r = K.gemm(a2d.cuda(), w2d.cuda(), bias=torch.from_numpy(b2).cuda(), alpha=alpha, lnr_y=p["half"], lnr_rows=rows, lnr_gamma=torch.from_numpy(gamma).cuda(),
                       lnr_beta=torch.from_numpy(beta).cuda(), out_f32=False, out_half=True, stats_out=bool(rng.integers(0, 2)))
            want = alpha * ln + a2.astype(np.float64) @ w2.astype(np.float64).T + b2
            e = rel(r["half"].float().cpu().numpy(), want) / (1.0e-3 if dt == "f16" else 6e-3)
            if "stats" in r:
                o = r["half"].float().cpu().numpy().astype(np.float64).reshape(M, N // 64, 64)
                assert np.allclose(r["stats"][..., 0].cpu().numpy(), o.sum(-1), rtol=2e-3, atol=0.5 if dt == "f16" else 4.0), "row statistics"
        os.environ.pop("AVEX_AMD_GEMM_GENERIC", None)
    elif mode == 0:
        r = K.gemm(ad, wd, bias=bd, variant=variant); e = rel(r["f32"].cpu().numpy(), ref) / 3e-6
    elif mode == 1:
        r = K.gemm(ad, wd, bias=bd, gelu=True, out_f32=False, out_half=True, variant=variant)
        e = rel(r["half"].float().cpu().numpy(), O.gelu_erf(ref.astype(np.float32))) / tolh
    elif mode == 2:
        r = K.gemm(ad, wd, bias=bd, resid_half=rhd, alpha=2.2133638, out_f32=False, out_half=True, variant=variant)
        e = rel(r["half"].float().cpu().numpy(), rh * 2.2133638 + ref) / tolh
    else:
        r = K.gemm(ad, wd, bias=bd, out_f32=False, out_half=True, variant=variant); e = rel(r["half"].float().cpu().numpy(), ref) / tolh
    worst["gemm"] = max(worst["gemm"], e)
    msg = f"case {c:3d}: gemm {dt} M={M} N={N} K={Kd} variant={variant} grid={os.environ.get('AVEX_AMD_GEMM_GRID', '-')} mode={mode} err/tol {e:.2f}"
    os.environ.pop("AVEX_AMD_GEMM_GRID", None)
    # ---------------- attention ----------------
    T = int(rng.choice([rng.integers(8, 100), rng.integers(100, 513), rng.integers(513, 1400)]))
    H = int(rng.choice([1, 4, 12])); B = int(rng.integers(1, 4))
    E = 64 * H
    qkv, qd = rnd_half(rng.standard_normal((B * T, 3 * E)).astype(np.float32), dt)
    use_bias, use_gate, use_pad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    tab = (0.5 * rng.standard_normal((H, 2 * T - 1))).astype(np.float32) if use_bias else None
    gw = (0.1 * rng.standard_normal((8, 64))).astype(np.float32); gb = (0.1 * rng.standard_normal(8)).astype(np.float32)
    ga = (1.0 + 0.2 * rng.standard_normal(H)).astype(np.float32)
    pad = None
    if use_pad:
        pad = np.zeros((B, T), bool)
        for b in range(B):
            pad[b, int(rng.integers(1, T + 1)):] = True
    if int(rng.integers(0, 3)) == 0:
        os.environ["AVEX_AMD_ATT_GRID"] = str(int(rng.choice([1, 3, 7])))
    else:
        os.environ.pop("AVEX_AMD_ATT_GRID", None)
    dev = lambda x: None if x is None else torch.from_numpy(x).cuda()
    out = K.attention(qd.cuda(), B, T, H, dev(tab), dev(gw) if use_bias and use_gate else None, dev(gb) if use_bias and use_gate else None,
                      dev(ga) if use_bias and use_gate else None, key_pad=dev(pad.astype(np.uint8)) if pad is not None else None).float().cpu().numpy()
    os.environ.pop("AVEX_AMD_ATT_GRID", None)
    x = qkv.reshape(B, T, 3, H, 64).astype(np.float64)
    q, k, v = x[:, :, 0].transpose(0, 2, 1, 3), x[:, :, 1].transpose(0, 2, 1, 3), x[:, :, 2].transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2) * 0.125
    if use_bias:
        idx = np.arange(T)[None, :] - np.arange(T)[:, None] + (T - 1)          # bias row index of (i, j): j - i + T - 1
        bias_m = tab[:, idx].astype(np.float64)                                  # [H, T, T]
        if use_gate:
            g8 = q @ gw.astype(np.float64).T + gb
            g2 = g8.reshape(B, H, T, 2, 4).sum(-1)
            sg = 1.0 / (1.0 + np.exp(-g2))
            gate = sg[..., 0:1] * (sg[..., 1:2] * ga.reshape(1, H, 1, 1) - 1.0) + 2.0
            s = s + gate * bias_m[None]
        else:
            s = s + bias_m[None]
    if pad is not None:
        s = np.where(pad[:, None, None, :], -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s); p /= p.sum(-1, keepdims=True)
    refo = (p @ v).transpose(0, 2, 1, 3).reshape(B * T, E)
    ea = rel(out, refo) / (2e-3 if dt == "f16" else 1.2e-2)
    worst["attention"] = max(worst["attention"], ea)
    # ---------------- attention for the other head widths (attention_hd.hip: the sequence probes' nn.MultiheadAttention) ----------------
    D = int(rng.choice([32, 64, 96, 128])); Th = int(rng.choice([rng.integers(1, 130), rng.integers(130, 700)])); Hh = int(rng.integers(1, 6))
    qkvh, qhd = rnd_half(rng.standard_normal((B * Th, 3 * Hh * D)).astype(np.float32), dt)
    padh = None
    if use_pad:
        padh = np.zeros((B, Th), bool)
        for b in range(B):
            padh[b, int(rng.integers(1, Th + 1)):] = True
    outh = K.attention_hd(qhd.cuda(), B, Th, Hh, D, key_pad=dev(padh.astype(np.uint8)) if padh is not None else None).float().cpu().numpy()
    xh = qkvh.reshape(B, Th, 3, Hh, D).astype(np.float64)
    qh, kh, vh = xh[:, :, 0].transpose(0, 2, 1, 3), xh[:, :, 1].transpose(0, 2, 1, 3), xh[:, :, 2].transpose(0, 2, 1, 3)
    sh = qh @ kh.transpose(0, 1, 3, 2) / np.sqrt(D)
    if padh is not None:
        sh = np.where(padh[:, None, None, :], -np.inf, sh)
    sh = sh - sh.max(-1, keepdims=True)
    ph = np.exp(sh); ph /= ph.sum(-1, keepdims=True)
    refh = (ph @ vh).transpose(0, 2, 1, 3).reshape(B * Th, Hh * D)
    eh = rel(outh, refh) / (2e-3 if dt == "f16" else 1.2e-2)
    worst["attention_hd"] = max(worst["attention_hd"], eh)
    print(msg + f" | attention T={T} H={H} B={B} bias={int(use_bias)} gate={int(use_gate)} pad={int(use_pad)} err/tol {ea:.2f}"
          + f" | attention_hd T={Th} H={Hh} D={D} err/tol {eh:.2f}", flush=True)
    if not (e < 1 and ea < 1 and eh < 1 and np.isfinite(out).all() and np.isfinite(outh).all()):
        print("VIOLATION"); sys.exit(1)
print(f"{cases} cases, worst error / tolerance: {worst}")
