"""GPU parity tests, one HIP kernel at a time, through the C ABI (`-m gpu`).

Each kernel is compared with the CPU oracle (oracle/beats_oracle.py, pinned against the real
reference by tests/test_oracle_golden.py) on the same seeded inputs.  MFMA kernels see inputs
rounded to the operand type, so the CPU side rounds the same inputs first: the remaining
difference is fp32 accumulation order (tolerances are written next to each check).
"""
import math
import os

import numpy as np
import pytest
import torch

from _util import max_abs, rel_l2, round_half
from avex_amd import synth
from oracle import beats_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DTYPES = ["f16", "bf16"]


def _dev(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda", dtype)


def _tdt(name):
    return torch.float16 if name == "f16" else torch.bfloat16


def test_eat_frontend_matches_oracle(built_lib):
    """EAT frontend (eat/audio_processor.py:72-143) through the C ABI vs the NumPy restatement: clip-mean removal, Hann
    window, no 2**15 scale, zero-padded rows after normalisation (5 s clip), truncation to 1024 frames (10.6 s clip)."""
    from avex_amd.eat_audio_processor import EATAudioProcessor
    proc = EATAudioProcessor()
    assert proc.hop_length == 160 and proc.target_length == 1024
    x = synth.noise_clips(3, 80000, seed=11) + np.float32(0.05)             # DC offset exercises mono - mono.mean()
    y = proc(torch.from_numpy(x).cuda())
    assert y.shape == (3, 1024, 128) and y.dtype == torch.float32 and y.is_cuda
    ref = O.eat_preprocess(x)
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-4                       # log-mel / 9.1: same tolerance class as the BEATs frontend
    pad = np.float32((0.0 + 4.268) / (2 * 4.569))
    assert np.allclose(y[:, 498:].cpu().numpy(), pad, atol=1e-6)
    long = synth.noise_clips(2, 170000, seed=12)
    yl = proc(long)                                                         # numpy in -> tensor on the input's (CPU) device, like the reference
    assert yl.shape == (2, 1024, 128) and not yl.is_cuda
    assert np.abs(yl.numpy() - O.eat_preprocess(long)).max() < 2e-4
    ps = EATAudioProcessor(norm_mean=0.0, norm_std=1.0)(torch.from_numpy(x).cuda()).cpu().numpy()   # per-sample statistics branch
    assert np.abs(ps - O.eat_preprocess(x, norm_mean=0.0, norm_std=1.0)).max() < 2e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_cast_roundtrip(built_lib, dtype):
    from avex_amd import kernels as K
    x = synth.normal("cast", (4099,), 3.0)
    h = K.to_half(_dev(x), dtype)
    assert h.dtype == _tdt(dtype)
    ref = round_half(x, dtype)
    assert np.array_equal(h.float().cpu().numpy(), ref)          # RNE cast is bit-exact
    assert np.array_equal(K.to_f32(h).cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("shape", [(300, 256, 192), (1984, 768, 768), (128, 128, 64), (77, 3072, 768), (2500, 512, 256), (1024, 768, 3072)])
def test_gemm_epilogues(built_lib, dtype, variant, shape):
    from avex_amd import kernels as K
    M, N, Kd = shape
    a = round_half(synth.normal(f"A{shape}", (M, Kd), 1.0), dtype)
    w = round_half(synth.normal(f"W{shape}", (N, Kd), 0.05), dtype)
    bias = synth.normal("bias", (N,), 0.1)
    resid = synth.normal("resid", (M, N), 1.0)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    ad, wd = _dev(a, _tdt(dtype)), _dev(w, _tdt(dtype))
    # plain + bias
    r = K.gemm(ad, wd, bias=_dev(bias), variant=variant)
    assert rel_l2(r["f32"].cpu().numpy(), ref) < 2e-6        # fp32 accumulate vs fp64
    # residual (DeepNorm) + raw tap + half copy
    alpha = 2.2133638
    r = K.gemm(ad, wd, bias=_dev(bias), resid=_dev(resid), alpha=alpha, out_half=True, out_raw=True, variant=variant)
    assert rel_l2(r["raw"].cpu().numpy(), ref) < 2e-6
    assert rel_l2(r["f32"].cpu().numpy(), resid * alpha + ref) < 2e-6
    assert np.array_equal(r["half"].float().cpu().numpy(), round_half(r["f32"].cpu().numpy(), dtype))
    rh = round_half(resid, dtype)
    r = K.gemm(ad, wd, bias=_dev(bias), resid_half=_dev(rh, _tdt(dtype)), alpha=alpha, variant=variant)
    assert rel_l2(r["f32"].cpu().numpy(), rh * alpha + ref) < 2e-6
    # half-only outputs take the branch-free epilogues of the 256-tile kernel: bias(+GELU) and bias+residual
    tolh = 6e-4 if dtype == "f16" else 5e-3      # one rounding to the operand type
    r = K.gemm(ad, wd, bias=_dev(bias), out_f32=False, out_half=True, variant=variant)
    assert rel_l2(r["half"].float().cpu().numpy(), ref) < tolh
    r = K.gemm(ad, wd, bias=_dev(bias), gelu=True, out_f32=False, out_half=True, variant=variant)
    assert rel_l2(r["half"].float().cpu().numpy(), O.gelu_erf(ref.astype(np.float32))) < tolh
    r = K.gemm(ad, wd, bias=_dev(bias), resid_half=_dev(rh, _tdt(dtype)), alpha=alpha, out_f32=False, out_half=True, variant=variant)
    assert np.array_equal(r["half"].float().cpu().numpy(), round_half((rh * alpha + ref).astype(np.float32), dtype)) or \
        rel_l2(r["half"].float().cpu().numpy(), rh * alpha + ref) < tolh
    # exact-erf GELU
    r = K.gemm(ad, wd, bias=_dev(bias), gelu=True, variant=variant)
    assert rel_l2(r["f32"].cpu().numpy(), O.gelu_erf(ref.astype(np.float32))) < 5e-6   # A&S 7.1.26 erf: |err| < 6e-7


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2500, 768, 256), (3000, 512, 128), (1100, 3072, 192)])
def test_gemm_persistent_walk_and_epilogues_agree(built_lib, dtype, shape, monkeypatch):
    """The 256-tile kernel is persistent: with the grid forced to 8 every workgroup walks many tiles (the continuous K stream across
    tile boundaries, the one-tile-ahead LDS-DMA of the bias row, the counted waits around the epilogue).  Every epilogue must give
    the same bits (i) whatever the grid, (ii) through its fast form and through the generic one (AVEX_AMD_GEMM_GENERIC=1), and
    (iii) for the plain / bias / GELU epilogues, as the 128-tile kernel (same K order, same arithmetic)."""
    from avex_amd import kernels as K
    M, N, Kd = shape
    a = _dev(round_half(synth.normal(f"pA{shape}", (M, Kd), 1.0), dtype), _tdt(dtype))
    w = _dev(round_half(synth.normal(f"pW{shape}", (N, Kd), 0.05), dtype), _tdt(dtype))
    bias = _dev(synth.normal("pbias", (N,), 0.1))
    rh = _dev(round_half(synth.normal("presid", (M, N), 1.0), dtype), _tdt(dtype))
    resid = _dev(synth.normal("presid32", (M, N), 1.0))
    cases = [dict(out_f32=False, out_half=True), dict(out_f32=False, out_half=True, gelu=True),
             dict(out_f32=False, out_half=True, resid_half=rh, alpha=2.2133638),
             dict(out_f32=False, out_half=True, resid_half=rh, alpha=2.2133638, stats_out=True),
             dict(out_f32=True, out_half=True, out_raw=True, resid=resid, alpha=2.2133638),
             dict(out_f32=True, resid_half=rh, alpha=0.5, gelu=True)]
    for ci, kw in enumerate(cases):
        monkeypatch.delenv("AVEX_AMD_GEMM_GRID", raising=False)
        monkeypatch.delenv("AVEX_AMD_GEMM_GENERIC", raising=False)
        ref = K.gemm(a, w, bias=bias, variant=5, **kw)
        monkeypatch.setenv("AVEX_AMD_GEMM_GRID", "8")
        for _ in range(2):          # twice: a stale prologue from the previous launch must not matter
            r8 = K.gemm(a, w, bias=bias, variant=5, **kw)
            for key in ref:
                assert torch.equal(r8[key], ref[key]), (ci, key, "grid 8")
        monkeypatch.setenv("AVEX_AMD_GEMM_GENERIC", "1")
        rg = K.gemm(a, w, bias=bias, variant=5, **kw)
        for key in ref:
            assert torch.equal(rg[key], ref[key]), (ci, key, "generic epilogue")
        monkeypatch.delenv("AVEX_AMD_GEMM_GRID", raising=False)
        monkeypatch.delenv("AVEX_AMD_GEMM_GENERIC", raising=False)
        if ci < 2:
            r3 = K.gemm(a, w, bias=bias, variant=3, **kw)
            assert torch.equal(r3["half"], ref["half"]), (ci, "128-tile kernel")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,grid", [(5, 496, 0), (3, 64, 0), (7, 100, 16), (2, 513, 0), (9, 65, 8), (1, 300, 0)])
def test_gemm_pooled_tap(built_lib, dtype, B, T, grid, monkeypatch):
    """A mean-pooled hook tap without the tap (avexhip_gemm_args.pool_part + avexhip_pool_reduce): per-clip means of acc + bias over
    clips of T rows, from per-64-row-block column sums split at the clip boundary.  Clip lengths that put the boundary everywhere in a
    block (64: never inside; 65, 100, 300, 496, 513), several tiles per workgroup, a residual and an f16 output alongside (the pooled
    value is the raw tap, before the residual), against the mean of the materialised tap and an fp64 product."""
    from avex_amd import kernels as K
    if grid:
        monkeypatch.setenv("AVEX_AMD_GEMM_GRID", str(grid))
    N, Kd = 512, 256
    M = B * T
    a = round_half(synth.normal(f"plA{M}", (M, Kd), 1.0), dtype)
    w = round_half(synth.normal("plW", (N, Kd), 0.06), dtype)
    bias = synth.normal("plb", (N,), 0.3)
    res = round_half(synth.normal(f"plR{M}", (M, N), 1.0), dtype)
    td = _tdt(dtype)
    r = K.gemm(_dev(a, td), _dev(w, td), bias=_dev(bias), resid_half=_dev(res, td), alpha=2.0, out_f32=False, out_half=True, pool_rows=T)
    raw = K.gemm(_dev(a, td), _dev(w, td), bias=_dev(bias), resid_half=_dev(res, td), alpha=2.0, out_f32=False, out_half=True, out_raw=True)
    assert torch.equal(r["half"], raw["half"])                          # the ordinary outputs do not notice
    pooled = r["pooled"].cpu().numpy()
    assert pooled.shape == (B, N) and np.isfinite(pooled).all()
    assert rel_l2(pooled, raw["raw"].cpu().numpy().reshape(B, T, N).mean(1)) < 2e-6
    want = (a.astype(np.float64) @ w.astype(np.float64).T + bias).reshape(B, T, N).mean(1)
    assert rel_l2(pooled, want) < 1e-5
    r2 = K.gemm(_dev(a, td), _dev(w, td), bias=_dev(bias), resid_half=_dev(res, td), alpha=2.0, out_f32=False, out_half=True, pool_rows=T)
    assert torch.equal(r2["pooled"], r["pooled"])                       # blocks are added in a fixed order: reproducible bits
    # the other two aggregations of extract_embeddings (beats_model.py:403-417): the maximum over a clip's rows and its first row -- exact
    tap = raw["raw"].reshape(B, T, N)
    rmax = K.gemm(_dev(a, td), _dev(w, td), bias=_dev(bias), resid_half=_dev(res, td), alpha=2.0, out_f32=False, out_half=True, pool_rows=T, pool_mode="max")
    assert torch.equal(rmax["pooled"], tap.max(dim=1)[0]) and torch.equal(rmax["half"], raw["half"])
    rcls = K.gemm(_dev(a, td), _dev(w, td), bias=_dev(bias), resid_half=_dev(res, td), alpha=2.0, out_f32=False, out_half=True, pool_rows=T, pool_mode="cls_token")
    assert torch.equal(rcls["pooled"], tap[:, 0]) and torch.equal(rcls["half"], raw["half"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Kd", [(128, 64), (64, 64), (64, 128), (128, 128), (256, 64), (64, 256), (128, 256), (256, 128),
                                  (64, 32), (128, 32), (96, 64), (64, 96), (128, 96), (160, 64), (64, 160), (128, 160)])
def test_gemm_skinny_streaming(built_lib, dtype, N, Kd):
    """The skinny streaming kernel (variant 7: the whole W in LDS, A rows straight into MFMA operand registers; EfficientNet's 1 x 1
    convolutions at the early stages) against the 128-tile kernel: same MFMA, same K order, same epilogue order -> the same bits.  A row
    count that is no multiple of 32, bias + half residual + SiLU, and the plain form; auto-selection from 32 768 rows."""
    from avex_amd import kernels as K
    M = 40000 + 13
    td = _tdt(dtype)
    a = _dev(round_half(synth.normal(f"skA{Kd}", (M, Kd), 1.0), dtype), td)
    w = _dev(round_half(synth.normal(f"skW{N}{Kd}", (N, Kd), 0.1), dtype), td)
    bias = _dev(synth.normal(f"skb{N}", (N,), 0.3))
    res = _dev(round_half(synth.normal(f"skR{N}", (M, N), 1.0), dtype), td)
    for kw in (dict(bias=bias, resid_half=res, alpha=0.5, silu=True), dict(bias=bias), dict()):
        r7 = K.gemm(a, w, out_f32=False, out_half=True, variant=7, **kw)
        if Kd % 64 == 0 and (N == 64 or N % 128 == 0):
            r0 = K.gemm(a, w, out_f32=False, out_half=True, **kw)["half"]      # auto: M >= 32 768 takes the skinny kernel
            assert torch.equal(r0, r7["half"])
        if N % 128 == 0 and Kd % 64 == 0:                                   # (the 128-tile kernel needs whole 128-column tiles and K-steps of 64)
            r3 = K.gemm(a, w, out_f32=False, out_half=True, variant=3, **kw)
            assert torch.equal(r7["half"], r3["half"])
    want = a[:2000].double() @ w.double().T
    got = K.gemm(a, w, out_f32=False, out_half=True, variant=7)["half"][:2000].double()
    assert float((got - want).norm() / want.norm()) < (2e-3 if dtype == "f16" else 1.2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,Kd", [(496, 768, 3072), (96, 768, 3072), (1000, 128, 1024), (300, 256, 2048)])
def test_gemm_split_k(built_lib, dtype, M, N, Kd):
    """Split-K of the 128-tile kernel for few-row, long-K products (one clip's fc2): fp32 partials added in order + the full epilogue
    (bias, half residual, raw tap, both outputs) in a second kernel, against the unsplit launch (same products, another summation order)
    and fp64."""
    from avex_amd import kernels as K
    td = _tdt(dtype)
    a = _dev(round_half(synth.normal(f"spA{M}", (M, Kd), 1.0), dtype), td)
    w = _dev(round_half(synth.normal(f"spW{N}", (N, Kd), 0.03), dtype), td)
    bias = _dev(synth.normal("spb", (N,), 0.3))
    res = _dev(round_half(synth.normal(f"spR{M}", (M, N), 1.0), dtype), td)
    kw = dict(bias=bias, resid_half=res, alpha=2.0, out_f32=True, out_half=True, out_raw=True, variant=3)
    r1 = K.gemm(a, w, **kw)
    r8 = K.gemm(a, w, splitk=True, **kw)
    for k in ("f32", "raw"):
        assert rel_l2(r8[k].cpu().numpy(), r1[k].cpu().numpy()) < 2e-6
    assert rel_l2(r8["half"].float().cpu().numpy(), r1["half"].float().cpu().numpy()) < (1e-3 if dtype == "f16" else 8e-3)
    want = (a.double() @ w.double().T + bias.double()).cpu().numpy()
    assert rel_l2(r8["raw"].cpu().numpy(), want) < 1e-5
    r8b = K.gemm(a, w, splitk=True, **kw)
    assert torch.equal(r8b["f32"], r8["f32"])          # partials are added in split order: reproducible


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,Kd", [(496, 768, 768), (496, 2304, 768), (130, 128, 64), (130, 128, 128), (700, 256, 192), (496, 768, 256),
                                    (1984, 768, 768), (496, 768, 3072)])
def test_gemm_lds_dma_equals_register_staging(built_lib, dtype, M, N, Kd):
    """The 128-tile kernel's LDS-DMA form (variant 3: what few-tile products take) and its register-staged form (variant 1) build the same
    LDS image and issue the same MFMAs in the same order: bit-identical, for one to 48 K-steps and partial row tiles; split-K against fp64.
    (A four-stage LDS pipeline for launches of fewer workgroups than CUs was tried here and measured level -- 24.5 against 22.4 us for one
    clip's out_proj: such launches are bound by dispatch, not by load latency -- and dropped.)"""
    from avex_amd import kernels as K
    td = _tdt(dtype)
    a = _dev(round_half(synth.normal(f"dpA{M}{Kd}", (M, Kd), 1.0), dtype), td)
    w = _dev(round_half(synth.normal(f"dpW{N}{Kd}", (N, Kd), 0.05), dtype), td)
    bias = _dev(synth.normal("dpb", (N,), 0.3))
    kw = dict(bias=bias, out_f32=True, out_half=True)
    r1 = K.gemm(a, w, variant=1, **kw)
    r3 = K.gemm(a, w, variant=3, **kw)
    assert torch.equal(r3["f32"], r1["f32"]) and torch.equal(r3["half"], r1["half"])
    want = (a.double() @ w.double().T + bias.double()).cpu().numpy()
    assert rel_l2(r3["f32"].cpu().numpy(), want) < 1e-5
    if Kd >= 1024:
        rs = K.gemm(a, w, variant=3, splitk=True, **kw)
        assert rel_l2(rs["f32"].cpu().numpy(), want) < 1e-5


def test_gemm_pooled_tap_refuses_short_clips(built_lib):
    from avex_amd import kernels as K
    from avex_amd._capi import AvexHipError
    a = torch.zeros((128, 256), dtype=torch.float16, device="cuda"); w = torch.zeros((256, 256), dtype=torch.float16, device="cuda")
    with pytest.raises(AvexHipError):
        K.gemm(a, w, out_f32=False, out_half=True, pool_rows=32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [700, 1536, 1539])
def test_gemm_folded_layernorm(built_lib, dtype, M, monkeypatch):
    """LayerNorm folded into the GEMMs around it (include/avexhip.h, avexhip_gemm_args): the producer writes raw rows y and
    per-segment statistics; a consumer takes LN(y) as its A operand through folded weights (W diag(gamma), bias + W beta,
    column sums); another takes LN(y) as its residual.  Checked against the explicit two-pass fp32 LayerNorm of the same
    rounded rows (backbone.py:363,374)."""
    from avex_amd import kernels as K
    E, F = 768, 1024
    td = _tdt(dtype)
    rnd = lambda a: round_half(a, dtype)
    a0 = rnd(synth.normal("lnA", (M, 256), 1.0)); w0 = rnd(synth.normal("lnW0", (E, 256), 0.08)); b0 = synth.normal("lnb0", (E,), 0.3)
    x = rnd(synth.normal("lnx", (M, E), 1.0))
    gamma = 1.0 + synth.normal("lng", (E,), 0.2); beta = synth.normal("lnbeta", (E,), 0.2)
    alpha = 2.2133638
    # producer: y = x * alpha + a0 @ w0.T + b0 (plain half residual), with statistics
    r = K.gemm(_dev(a0, td), _dev(w0, td), bias=_dev(b0), resid_half=_dev(x, td), alpha=alpha, out_f32=False, out_half=True, stats_out=True)
    y = r["half"].float().cpu().numpy()
    y_ref = (x * alpha + a0.astype(np.float64) @ w0.astype(np.float64).T + b0)
    assert rel_l2(y, y_ref) < (6e-4 if dtype == "f16" else 5e-3)
    st = r["stats"].cpu().numpy()
    seg = y.reshape(M, E // 64, 64).astype(np.float64)
    # (sums are taken from the fp32 values before the row is rounded to the operand type)
    assert np.allclose(st[..., 0], seg.sum(-1), rtol=1e-3, atol=0.05 if dtype == "f16" else 0.4) and np.allclose(st[..., 1], (seg ** 2).sum(-1), rtol=2e-3 if dtype == "f16" else 1e-2, atol=0.3)
    ln = O.layer_norm(y.astype(np.float32), gamma.astype(np.float32), beta.astype(np.float32)).astype(np.float64)   # LN of the ROUNDED rows
    rows = K.ln_rowstats(r["stats"])             # (rstd, -mean rstd) per row, from the producer's partial sums
    yd = y.astype(np.float64)
    # (the sums are of the fp32 values; yd is what was stored, one rounding to the operand type later)
    assert np.allclose(rows[:M, 0].cpu().numpy(), 1.0 / np.sqrt(yd.var(1) + 1e-5), rtol=2e-4 if dtype == "f16" else 2e-3)
    assert np.allclose(rows[:M, 1].cpu().numpy(), -yd.mean(1) / np.sqrt(yd.var(1) + 1e-5), rtol=2e-3, atol=2e-4 if dtype == "f16" else 2e-3)
    # consumer A: gelu(LN(y) @ w1.T + b1) through folded weights
    w1 = synth.normal("lnW1", (F, E), 0.05); b1 = synth.normal("lnb1", (F,), 0.1)
    w1f = rnd((w1 * gamma[None, :]).astype(np.float32))
    s1 = w1f.astype(np.float64).sum(1).astype(np.float32)
    b1f = (b1 + w1.astype(np.float64) @ beta).astype(np.float32)
    for gelu in (False, True):
        r1 = K.gemm(r["half"], _dev(w1f, td), bias=_dev(b1f), gelu=gelu, out_f32=False, out_half=True, ln_rows=rows, ln_s=_dev(s1))
        # exact model of the fold: LN(y) (gamma folded and ROUNDED into w1f) @ ...
        mu = y.astype(np.float64).mean(1, keepdims=True); rstd = 1.0 / np.sqrt(y.astype(np.float64).var(1, keepdims=True) + 1e-5)
        ref1 = ((y - mu) * rstd) @ w1f.astype(np.float64).T + b1f
        if gelu:
            ref1 = O.gelu_erf(ref1.astype(np.float32))
        assert rel_l2(r1["half"].float().cpu().numpy(), ref1) < (6e-4 if dtype == "f16" else 5e-3)
        # the fast epilogue (row pairs by LDS-DMA, two rows per lane: an odd M ends in half a pair) against the generic one (plain loads)
        monkeypatch.setenv("AVEX_AMD_GEMM_GENERIC", "1")
        r1x = K.gemm(r["half"], _dev(w1f, td), bias=_dev(b1f), gelu=gelu, out_f32=False, out_half=True, ln_rows=rows, ln_s=_dev(s1))
        monkeypatch.delenv("AVEX_AMD_GEMM_GENERIC")
        assert torch.equal(r1x["half"], r1["half"])
        assert rel_l2(r1["half"][-1].float().cpu().numpy(), ref1[-1]) < (2e-3 if dtype == "f16" else 1.5e-2)       # the last row on its own
        # and it agrees with the unfused formulation up to the rounding of the folded weights
        ref_unf = ln @ w1.astype(np.float64).T + b1
        if gelu:
            ref_unf = O.gelu_erf(ref_unf.astype(np.float32))
        assert rel_l2(r1["half"].float().cpu().numpy(), ref_unf) < (1.2e-3 if dtype == "f16" else 8e-3)
    # generic epilogue with the fold (fp32 output)
    r1g = K.gemm(r["half"], _dev(w1f, td), bias=_dev(b1f), ln_rows=rows, ln_s=_dev(s1))
    # (the statistics come from the fp32 rows before rounding: mu differs from the rounded rows' mean by ~1e-5 sigma)
    assert rel_l2(r1g["f32"].cpu().numpy(), ((y - mu) * rstd) @ w1f.astype(np.float64).T + b1f) < (1e-4 if dtype == "f16" else 1e-3)
    # consumer R: out = alpha * LN(y) + a2 @ w2.T + b2, fast path (+ statistics), and generic path with a raw tap
    a2 = rnd(synth.normal("lnA2", (M, 512), 1.0)); w2 = rnd(synth.normal("lnW2", (E, 512), 0.05)); b2 = synth.normal("lnb2", (E,), 0.1)
    raw_ref = a2.astype(np.float64) @ w2.astype(np.float64).T + b2
    ref2 = alpha * ln + raw_ref
    kw = dict(bias=_dev(b2), alpha=alpha, lnr_y=r["half"], lnr_rows=rows, lnr_gamma=_dev(gamma.astype(np.float32)), lnr_beta=_dev(beta.astype(np.float32)))
    r2 = K.gemm(_dev(a2, td), _dev(w2, td), out_f32=False, out_half=True, stats_out=True, **kw)
    out2 = r2["half"].float().cpu().numpy()
    assert rel_l2(out2, ref2) < (6e-4 if dtype == "f16" else 5e-3)
    seg2 = out2.reshape(M, E // 64, 64).astype(np.float64)
    assert np.allclose(r2["stats"][..., 0].cpu().numpy(), seg2.sum(-1), rtol=1e-3, atol=0.05 if dtype == "f16" else 0.4)
    assert np.allclose(r2["stats"][..., 1].cpu().numpy(), (seg2 ** 2).sum(-1), rtol=2e-3 if dtype == "f16" else 1e-2, atol=0.3)
    r3 = K.gemm(_dev(a2, td), _dev(w2, td), out_f32=False, out_half=True, out_raw=True, stats_out=True, **kw)
    assert torch.equal(r3["half"], r2["half"])                  # the tap-carrying epilogue repeats the branch-free one operation for operation
    assert rel_l2(r3["raw"].cpu().numpy(), raw_ref) < 2e-6
    assert torch.equal(r3["stats"], r2["stats"])
    r4 = K.gemm(_dev(a2, td), _dev(w2, td), out_f32=True, **kw)
    assert rel_l2(r4["f32"].cpu().numpy(), ref2) < (2e-5 if dtype == "f16" else 5e-4)


def test_gemm_rejects_bad_shapes(built_lib):
    from avex_amd import kernels as K
    from avex_amd._capi import AvexHipError
    a = torch.zeros((8, 64), dtype=torch.float16, device="cuda")
    with pytest.raises(AvexHipError):
        K.gemm(a, torch.zeros((100, 64), dtype=torch.float16, device="cuda"))    # N % 128
    with pytest.raises(AvexHipError):
        K.gemm(torch.zeros((8, 40), dtype=torch.float16, device="cuda"),
               torch.zeros((128, 40), dtype=torch.float16, device="cuda"))       # K % 64


@pytest.mark.parametrize("C", [768, 512, 64])
def test_layernorm(built_lib, C):
    from avex_amd import kernels as K
    x = synth.normal(f"ln{C}", (333, C), 2.0) + 0.5
    w = 1.0 + synth.normal("lnw", (C,), 0.1)
    b = synth.normal("lnb", (C,), 0.1)
    o32, oh = K.layernorm(_dev(x), _dev(w), _dev(b))
    ref = O.layer_norm(x, w.astype(np.float32), b.astype(np.float32))
    assert max_abs(o32.cpu().numpy(), ref) < 5e-6
    assert np.array_equal(oh.float().cpu().numpy(), round_half(o32.cpu().numpy(), "f16"))
    # half input (residual stream kept in the operand type)
    xh = round_half(x, "f16")
    o32h, _ = K.layernorm(_dev(xh, torch.float16), _dev(w), _dev(b))
    assert max_abs(o32h.cpu().numpy(), O.layer_norm(xh, w.astype(np.float32), b.astype(np.float32))) < 5e-6


def test_mean_pool(built_lib):
    from avex_amd import kernels as K
    x = synth.normal("pool", (3, 496, 768), 1.0)
    out = K.mean_pool(_dev(x)).cpu().numpy()
    assert max_abs(out, x.mean(axis=1, dtype=np.float64)) < 1e-6


def _attention_ref(qkv, B, T, H, table, gw, gb, ga, key_pad=None):
    E = H * 64
    q, k, v = [qkv[:, i * E:(i + 1) * E].reshape(B, T, H, 64).transpose(0, 2, 1, 3).astype(np.float64) for i in range(3)]
    s = q @ k.transpose(0, 1, 3, 2) * 0.125
    if table is not None:
        bias = O.position_bias(table, T, table.shape[0], 800 if table.shape[0] == 320 else 64).astype(np.float64)
        if gw is not None:
            g8 = q @ gw.T.astype(np.float64) + gb
            g2 = g8.reshape(B, H, T, 2, 4).sum(-1)
            sg = 1.0 / (1.0 + np.exp(-g2))
            gate = sg[..., 0:1] * (sg[..., 1:2] * ga.reshape(1, H, 1, 1) - 1.0) + 2.0
            s = s + gate * bias[None]
        else:
            s = s + bias[None]
    if key_pad is not None:
        s = np.where(key_pad[:, None, None, :], -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    o = (e / e.sum(-1, keepdims=True)) @ v
    return o.transpose(0, 2, 1, 3).reshape(B * T, E)


def _toeplitz(table, T, nb, md):
    from avex_amd import kernels as K
    H = table.shape[1]
    tab = np.empty((H, 2 * T - 1), np.float32)
    for r in range(2 * T - 1):
        tab[:, r] = table[K.rel_bucket(r - (T - 1), nb, md)]
    return tab


@pytest.mark.parametrize("variant", ["1", "2", "3"])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [496, 48, 384, 33, 512, 1, 17, 257, 300])
def test_attention(built_lib, dtype, T, variant, monkeypatch):
    from avex_amd import kernels as K
    monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", variant)
    B, H = 2, 12
    E = H * 64
    qkv = round_half(synth.normal(f"qkv{T}", (B * T, 3 * E), 1.0), dtype)
    table = synth.normal("rel", (320, H), 0.5)
    gw = synth.normal("gw", (8, 64), 0.1); gb = synth.normal("gb", (8,), 0.1); ga = 1.0 + synth.normal("ga", (H,), 0.2)
    tab = _toeplitz(table, T, 320, 800)
    out = K.attention(_dev(qkv, _tdt(dtype)), B, T, H, _dev(tab), _dev(gw), _dev(gb), _dev(ga))
    ref = _attention_ref(qkv, B, T, H, table, gw, gb, ga)
    # P is rounded to the operand type before P@V and the output is stored in it:
    # f16 ~ 2^-11 per element, bf16 ~ 2^-8
    tol = 1.5e-3 if dtype == "f16" else 1.2e-2
    assert rel_l2(out.float().cpu().numpy(), ref) < tol


def _plain_attention_ref(qkv, B, T, H, D, key_pad=None):
    E = H * D
    q, k, v = [qkv[:, i * E:(i + 1) * E].reshape(B, T, H, D).transpose(0, 2, 1, 3).astype(np.float64) for i in range(3)]
    s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(D)
    if key_pad is not None:
        s = np.where(key_pad[:, None, None, :], -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    o = (e / e.sum(-1, keepdims=True)) @ v
    return o.transpose(0, 2, 1, 3).reshape(B * T, E)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D,H", [(96, 8), (32, 4), (128, 3), (64, 5)])
@pytest.mark.parametrize("T", [496, 37, 128, 300, 513, 1])
def test_attention_other_head_widths(built_lib, dtype, D, H, T):
    """attention_hd.hip (the sequence probes' torch.nn.MultiheadAttention: 8 heads of 96 in the shipped configs) against fp64 on the same
    rounded operands: chunk boundaries (128 keys), partial key tiles, more than one query block, a key padding mask."""
    from avex_amd import kernels as K
    B = 3
    qkv = round_half(synth.normal(f"qkvhd{T}{D}", (B * T, 3 * H * D), 1.0), dtype)
    tol = 1.5e-3 if dtype == "f16" else 1.2e-2
    plain = K.attention_hd(_dev(qkv, _tdt(dtype)), B, T, H, D).float().cpu().numpy()
    assert rel_l2(plain, _plain_attention_ref(qkv, B, T, H, D)) < tol
    if T > 1:
        pad = np.zeros((B, T), bool); pad[1, T // 2:] = True; pad[2, :min(3, T - 1)] = True
        out = K.attention_hd(_dev(qkv, _tdt(dtype)), B, T, H, D, key_pad=_dev(pad.astype(np.uint8), torch.uint8))
        assert rel_l2(out.float().cpu().numpy(), _plain_attention_ref(qkv, B, T, H, D, key_pad=pad)) < tol
    if D == 64:      # the encoders' kernel without a bias table computes the same thing
        enc = K.attention(_dev(qkv, _tdt(dtype)), B, T, H, None, None, None, None)
        assert rel_l2(plain, enc.float().cpu().numpy()) < tol


def test_attention_other_head_widths_large_logits(built_lib):
    """Scores far outside the softmax's comfortable range (|s| ~ 300) and one dominating key per row: no overflow, no NaN."""
    from avex_amd import kernels as K
    B, T, H, D = 2, 260, 2, 96
    qkv = round_half(synth.normal("qkvhdbig", (B * T, 3 * H * D), 1.0), "f16")
    qkv[:, :2 * H * D] *= 6.0
    qkv = round_half(qkv, "f16")
    out = K.attention_hd(_dev(qkv, torch.float16), B, T, H, D).float().cpu().numpy()
    assert np.isfinite(out).all() and rel_l2(out, _plain_attention_ref(qkv, B, T, H, D)) < 3e-3


def test_attention_key_padding_and_plain_bias(built_lib):
    from avex_amd import kernels as K
    B, H, T = 2, 12, 96
    E = H * 64
    qkv = round_half(synth.normal("qkvpad", (B * T, 3 * E), 1.0), "f16")
    table = synth.normal("rel", (320, H), 0.5)
    tab = _toeplitz(table, T, 320, 800)
    pad = np.zeros((B, T), bool); pad[1, 40:] = True
    out = K.attention(_dev(qkv, torch.float16), B, T, H, _dev(tab), None, None, None,
                      key_pad=_dev(pad.astype(np.uint8), torch.uint8))
    ref = _attention_ref(qkv, B, T, H, table, None, None, None, key_pad=pad)
    assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,grid", [(499, 0), (499, 5), (256, 3), (257, 2), (40, 9)])
def test_attention_without_a_bias_table(built_lib, T, grid, dtype, monkeypatch):
    """The encoders without relative position bias (AVES / wav2vec2 at 499 frames, EAT below 513 tokens) take the default kernel's BIAS = false
    instantiation (variant 3 up to 512 tokens): plain softmax(q k^T / 8) v, with and without a key padding mask, workgroups that walk several
    items, against fp64 -- and against variant 2 on the same inputs."""
    from avex_amd import kernels as K
    if grid:
        monkeypatch.setenv("AVEX_AMD_ATT_GRID", str(grid))
    B, H = 3, 12
    E = H * 64
    qkv = round_half(synth.normal(f"qkvnb{T}", (B * T, 3 * E), 1.0), dtype)
    pad = np.zeros((B, T), bool); pad[1, T // 2:] = True; pad[2, :min(35, T - 1)] = True      # incl. a fully masked first key tile
    tol = 1.5e-3 if dtype == "f16" else 1.2e-2
    outs = {}
    for variant in ("3", "2"):
        monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", variant)
        o1 = K.attention(_dev(qkv, _tdt(dtype)), B, T, H, None, None, None, None).float().cpu().numpy()
        o2 = K.attention(_dev(qkv, _tdt(dtype)), B, T, H, None, None, None, None, key_pad=_dev(pad.astype(np.uint8), torch.uint8)).float().cpu().numpy()
        outs[variant] = (o1, o2)
        assert rel_l2(o1, _attention_ref(qkv, B, T, H, None, None, None, None)) < tol
        assert rel_l2(o2, _attention_ref(qkv, B, T, H, None, None, None, None, key_pad=pad)) < tol
    assert rel_l2(outs["3"][0], outs["2"][0]) < tol and rel_l2(outs["3"][1], outs["2"][1]) < tol


_ROW_CASES = [(1000, 768, 0), (128, 768, 0), (129, 128, 0), (5000, 768, 3), (2050, 3072, 5), (40000, 768, 0)]


@pytest.mark.parametrize("dtype,M,Kd,grid", [(d,) + c for d in DTYPES for c in _ROW_CASES] + [("f16", 126976, 768, 0)])      # (the full size once)
def test_gemm_full_row_kernel_is_bit_identical(built_lib, dtype, M, Kd, grid, monkeypatch):
    """gemm_row.hip (variant 8: one workgroup = 128 rows x all 768 columns; the attention output projection, backbone.py:572 + :360-362)
    against the streaming kernel's residual epilogue (variant 5) on the same inputs: outputs, partial statistics and the finished row
    statistics BIT FOR BIT (same MFMA chain over k, same epilogue operations, ln_rowstats_kernel's additions), in both residual forms
    (plain half residual, LayerNorm(lnr_y) on the fly), ragged last tiles, several tiles per workgroup, K = 128 / 768 / 3072; and the
    streaming kernel's result against fp64 (its own tests do that in depth)."""
    from avex_amd import kernels as K
    if grid:
        monkeypatch.setenv("AVEX_AMD_GEMM_GRID", str(8 * grid))      # (the streaming kernel's grid is a multiple of 8; the row kernel takes min(grid, tiles))
    E = 768
    td = _tdt(dtype)
    g = torch.Generator().manual_seed(M + Kd)
    a = (torch.randn(M, Kd, generator=g)).to(td).cuda()
    w = (torch.randn(E, Kd, generator=g) * Kd ** -0.5).to(td).cuda()
    x = torch.randn(M, E, generator=g).to(td).cuda()
    bias = torch.randn(E, generator=g).cuda()
    gamma = (1.0 + 0.2 * torch.randn(E, generator=g)).cuda(); beta = (0.2 * torch.randn(E, generator=g)).cuda()
    alpha = 2.2133638
    # (rstd, -mu rstd) of the residual rows, as a producer would have left them
    xf = x.float()
    rstd = 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)
    lrows = torch.zeros(M + (M & 1), 2, device="cuda"); lrows[:M, 0] = rstd; lrows[:M, 1] = -xf.mean(1) * rstd
    forms = {"plain": dict(resid_half=x), "lnr": dict(lnr_y=x, lnr_rows=lrows, lnr_gamma=gamma, lnr_beta=beta)}
    for name, kw in forms.items():
        out = {}
        for v in (5, 8):
            out[v] = K.gemm(a, w, bias=bias, alpha=alpha, out_f32=False, out_half=True, stats_out=True, rows_eps=1e-5, variant=v, **kw)
        assert torch.equal(out[8]["half"], out[5]["half"]), (name, "output")
        assert torch.equal(out[8]["stats"], out[5]["stats"]), (name, "partial statistics")
        assert torch.equal(out[8]["rows"][:M], out[5]["rows"][:M]), (name, "row statistics")
        assert torch.equal(out[5]["rows"][:M], K.ln_rowstats(out[5]["stats"], 1e-5)[:M])
        # without the partial statistics (what the encoder asks for) and without any
        r8 = K.gemm(a, w, bias=bias, alpha=alpha, out_f32=False, out_half=True, rows_eps=1e-5, variant=8, **kw)
        assert torch.equal(r8["half"], out[5]["half"]) and torch.equal(r8["rows"][:M], out[5]["rows"][:M])
        r8 = K.gemm(a, w, bias=bias, alpha=alpha, out_f32=False, out_half=True, variant=8, **kw)
        assert torch.equal(r8["half"], out[5]["half"])
        if M <= 5000:
            res = xf.double() * alpha if name == "plain" else alpha * (torch.nn.functional.layer_norm(xf, (E,), gamma, beta, 1e-5)).double()
            ref = (res + a.double() @ w.double().T + bias.double()).cpu().numpy()
            assert rel_l2(out[8]["half"].float().cpu().numpy(), ref) < (6e-4 if dtype == "f16" else 5e-3), name
    # the automatic choice (variant 0) with AVEX_AMD_GEMM_ROW=1 takes the row kernel from 32 768 rows at K <= 1024: same bits either way, so only the plumbing shows
    if M >= 32768:
        monkeypatch.setenv("AVEX_AMD_GEMM_ROW", "1")
        r0 = K.gemm(a, w, bias=bias, alpha=alpha, out_f32=False, out_half=True, rows_eps=1e-5, **forms["lnr"])
        assert torch.equal(r0["half"], out[5]["half"]) and torch.equal(r0["rows"][:M], out[5]["rows"][:M])
    with pytest.raises(K.AvexHipError, match="variant 8"):
        K.gemm(a, w, bias=bias, out_f32=True, out_half=True, resid_half=x, variant=8)          # an fp32 output is not its business


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_default_kernel_cross_checked_against_variant_2(built_lib, dtype, monkeypatch):
    """The CI cross-check of the default kernel (variant 3: hand-counted vmcnt waits, inline-asm Q loads, deferred softmax reference; default
    up to 512 tokens and, without a bias table, for the main block of 513 .. 544) against the kernel it replaced (variant 2, compiler-
    scheduled waits) over RAGGED lengths -- every tile boundary of either kernel -- with the gated bias, a key padding mask whose padded
    range covers whole key tiles, and a clip with every key padded (both kernels must treat the all-masked rows the same way)."""
    from avex_amd import kernels as K
    B, H = 3, 4
    E = H * 64
    table = synth.normal("relx", (320, H), 0.5)
    gw = synth.normal("gwx", (8, 64), 0.1); gb = synth.normal("gbx", (8,), 0.1); ga = 1.0 + synth.normal("gax", (H,), 0.2)
    tol = 1.5e-3 if dtype == "f16" else 1.2e-2
    worst = 0.0
    for T in (1, 2, 15, 16, 17, 31, 32, 33, 47, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 479, 480, 481, 495, 496, 497, 511, 512, 513, 514, 520, 543, 544):
        qkv = _dev(round_half(synth.normal(f"qkvx{T}", (B * T, 3 * E), 1.0), dtype), _tdt(dtype))
        pad = np.zeros((B, T), bool); pad[1, max(1, T // 3):] = True; pad[2, :] = True      # clip 1: a padded tail (whole key tiles of it), clip 2: every key padded
        padd = _dev(pad.astype(np.uint8), torch.uint8)
        tab = _dev(_toeplitz(table, T, 320, 800)) if T <= 512 else None      # beyond 512 tokens only the bias-free form runs on variant 3
        gate = (_dev(gw), _dev(gb), _dev(ga)) if tab is not None else (None, None, None)
        outs = {}
        for variant in ("3", "2"):
            monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", variant)
            outs[variant] = (K.attention(qkv, B, T, H, tab, *gate).float().cpu().numpy(),
                             K.attention(qkv, B, T, H, tab, *gate, key_pad=padd).float().cpu().numpy().reshape(B, T, E))
        assert np.isfinite(outs["3"][0]).all() and rel_l2(outs["3"][0], outs["2"][0]) < tol, T
        a, b = outs["3"][1], outs["2"][1]
        assert np.isfinite(a[:2]).all() and rel_l2(a[:2], b[:2]) < tol, T
        assert np.array_equal(np.isnan(a[2]), np.isnan(b[2])), T                      # the all-masked clip: NaN where the other kernel has NaN ...
        assert np.allclose(np.nan_to_num(a[2]), np.nan_to_num(b[2]), atol=2e-2), T   # ... and the same numbers elsewhere
        worst = max(worst, rel_l2(outs["3"][0], outs["2"][0]))
    print(f"variant 3 vs variant 2, {dtype}: worst rel-L2 over the ragged lengths {worst:.2e}")


@pytest.mark.parametrize("variant", ["1", "2", "3"])
@pytest.mark.parametrize("T,grid", [(496, 7), (200, 5), (300, 60), (512, 1)])
def test_attention_persistent_items(built_lib, T, grid, variant, monkeypatch):
    """The persistent attention (variant 2) walks several (head, clip) items per workgroup: ranges that cross a head
    seam, a ragged last workgroup, per-clip key padding, one- and two-half key ranges.  Variant 1 runs the same cases."""
    from avex_amd import kernels as K
    monkeypatch.setenv("AVEX_AMD_ATT_GRID", str(grid))
    monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", variant)
    B, H = 5, 12
    E = H * 64
    qkv = round_half(synth.normal(f"qkvp{T}", (B * T, 3 * E), 1.0), "f16")
    table = synth.normal("rel", (320, H), 0.5)
    gw = synth.normal("gw", (8, 64), 0.1); gb = synth.normal("gb", (8,), 0.1); ga = 1.0 + synth.normal("ga", (H,), 0.2)
    tab = _toeplitz(table, T, 320, 800)
    pad = np.zeros((B, T), bool); pad[1, T // 3:] = True; pad[3, T - 5:] = True; pad[4, :40] = True   # incl. a masked FIRST key tile
    out = K.attention(_dev(qkv, torch.float16), B, T, H, _dev(tab), _dev(gw), _dev(gb), _dev(ga),
                      key_pad=_dev(pad.astype(np.uint8), torch.uint8))
    ref = _attention_ref(qkv, B, T, H, table, gw, gb, ga, key_pad=pad)
    assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3
    out2 = K.attention(_dev(qkv, torch.float16), B, T, H, _dev(tab), _dev(gw), _dev(gb), _dev(ga))
    ref2 = _attention_ref(qkv, B, T, H, table, gw, gb, ga)
    assert rel_l2(out2.float().cpu().numpy(), ref2) < 1.5e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", ["1", "2", "3"])
def test_attention_large_logits(built_lib, variant, dtype, monkeypatch):
    """Row maxima that keep growing along the keys exercise the deferred-maximum rescale of variant 2 (the branch is
    data dependent and rare on gaussian scores): q.k grows with the key index, so every tile moves the reference."""
    from avex_amd import kernels as K
    monkeypatch.setenv("AVEX_AMD_ATT_GRID", "3")
    monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", variant)
    B, H, T = 2, 12, 496
    E = H * 64
    rng = np.random.default_rng(5)
    qkv = synth.normal("qkvbig", (B * T, 3 * E), 1.0).reshape(B, T, 3, H, 64)
    ramp = (np.arange(T, dtype=np.float32) / T)[None, :, None, None]
    qkv[:, :, 0] = 3.0 + 0.3 * qkv[:, :, 0]                       # q ~ 3
    qkv[:, :, 1] = 3.0 * ramp * (1.0 + 0.1 * qkv[:, :, 1])        # k grows along the sequence: logits up to ~ 70 (log2 ~ 100)
    qkv = round_half(qkv.reshape(B * T, 3 * E), dtype)
    table = synth.normal("rel", (320, H), 0.5)
    tab = _toeplitz(table, T, 320, 800)
    out = K.attention(_dev(qkv, _tdt(dtype)), B, T, H, _dev(tab), None, None, None)
    ref = _attention_ref(qkv, B, T, H, table, None, None, None)
    assert np.isfinite(out.float().cpu().numpy()).all()
    # bf16 through the PUBLIC entry point (plain Q): the kernel scales Q by log2(e) / 8 rounded to bf16 (0.18 % off) and rounds the product
    # again, so logits of ~ 70 move by ~ 0.1 -- the bar below states that; the handles fold log2(e) into W_q in fp32 instead (q_log2e)
    err = rel_l2(out.float().cpu().numpy(), ref)
    print(f"large logits {dtype} variant {variant}: {err:.3e}")
    assert err < (1.5e-3 if dtype == "f16" else 5e-2)


@pytest.mark.parametrize("T,grid", [(513, 0), (600, 7), (1000, 5), (1537, 3), (1537, 7), (2100, 11), (520, 0), (544, 4), (545, 0)])
def test_attention_long_clips(built_lib, T, grid, monkeypatch):
    """More than 512 tokens (the reference has no length limit, backbone.py:151-221; EAT has 513): queries in blocks of 512,
    keys in blocks of 256, a bias-row window and a key mask per (query block, key block) phase.  Gate, key padding (a masked
    first key tile, a clip whose last query block is one row), items that cross a head seam, workgroups whose run of (item, query
    block) units starts in the middle of an item (grid 5 at T = 1000: 72 units in runs of 15), vs the fp64 restatement."""
    from avex_amd import kernels as K
    if grid:
        monkeypatch.setenv("AVEX_AMD_ATT_GRID", str(grid))
    B, H = 3, 12
    E = H * 64
    qkv = round_half(synth.normal(f"qkvL{T}", (B * T, 3 * E), 1.0), "f16")
    table = synth.normal("rel", (320, H), 0.5)
    gw = synth.normal("gw", (8, 64), 0.1); gb = synth.normal("gb", (8,), 0.1); ga = 1.0 + synth.normal("ga", (H,), 0.2)
    tab = _toeplitz(table, T, 320, 800)
    out = K.attention(_dev(qkv, torch.float16), B, T, H, _dev(tab), _dev(gw), _dev(gb), _dev(ga))
    ref = _attention_ref(qkv, B, T, H, table, gw, gb, ga)
    assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3
    pad = np.zeros((B, T), bool); pad[1, T // 3:] = True; pad[2, :40] = True; pad[0, T - 3:] = True
    out = K.attention(_dev(qkv, torch.float16), B, T, H, _dev(tab), _dev(gw), _dev(gb), _dev(ga), key_pad=_dev(pad.astype(np.uint8), torch.uint8))
    ref = _attention_ref(qkv, B, T, H, table, gw, gb, ga, key_pad=pad)
    assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3
    # no bias table at all (EAT / AVES): plain softmax(q k^T / 8) v
    out = K.attention(_dev(qkv, torch.float16), B, T, H, None, None, None, None)
    ref = _attention_ref(qkv, B, T, H, None, None, None, None)
    assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3
    if 0 < T % 512 <= 32:
        # a last query block of at most 8 rows (32 when asked) runs in the one-wave-per-row tail kernel: same rows through the streamed kernel instead
        monkeypatch.setenv("AVEX_AMD_ATT_TAIL_ROWS", "32")
        out = K.attention(_dev(qkv, torch.float16), B, T, H, None, None, None, None)
        assert rel_l2(out.float().cpu().numpy(), ref) < 1.5e-3
        monkeypatch.delenv("AVEX_AMD_ATT_TAIL_ROWS")
        monkeypatch.setenv("AVEX_AMD_ATT_NO_TAIL", "1")
        out2 = K.attention(_dev(qkv, torch.float16), B, T, H, None, None, None, None)
        monkeypatch.delenv("AVEX_AMD_ATT_NO_TAIL")
        o1, o2 = out.float().cpu().numpy().reshape(B, T, E), out2.float().cpu().numpy().reshape(B, T, E)
        # With the tail, the main block of a clip of up to 544 tokens runs on variant 3's nine-tile form (round 5), without it on variant 2's
        # query blocks: two kernels, agreement to the tolerance; beyond 544 tokens both paths are variant 2 and the main blocks are bit-identical
        if T <= 544:
            assert rel_l2(o1[:, :T - T % 512], o2[:, :T - T % 512]) < 1.5e-3
            monkeypatch.setenv("AVEX_AMD_ATT_VARIANT", "2")                           # the same split on variant 2: bit for bit
            monkeypatch.setenv("AVEX_AMD_ATT_TAIL_ROWS", "32")
            o3 = K.attention(_dev(qkv, torch.float16), B, T, H, None, None, None, None).float().cpu().numpy().reshape(B, T, E)
            monkeypatch.delenv("AVEX_AMD_ATT_VARIANT"); monkeypatch.delenv("AVEX_AMD_ATT_TAIL_ROWS")
            assert np.array_equal(o3[:, :T - T % 512], o2[:, :T - T % 512])
        else:
            assert np.array_equal(o1[:, :T - T % 512], o2[:, :T - T % 512])             # the main blocks are untouched by the split
        assert rel_l2(o1[:, T - T % 512:], o2[:, T - T % 512:]) < 1.5e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T", [496, 48, 130, 513, 1100])
def test_posconv(built_lib, dtype, T):
    from avex_amd import kernels as K
    B, E, G, Kt = 2, 768, 16, 128
    x = synth.normal(f"pcx{T}", (B, T, E), 1.0)
    v = synth.normal("pcv", (E, E // G, Kt), math.sqrt(4.0 / (Kt * E)))
    g = (np.sqrt((v.astype(np.float64) ** 2).sum((0, 1), keepdims=True)) * (1.0 + synth.normal("pcg", (1, 1, Kt), 0.1))).astype(np.float32)
    bias = synth.normal("pcb", (E,), 0.05)
    wp = K.posconv_pack(_dev(g), _dev(v), G, dtype)
    # packed layout [g][u][v][g4][o][8] (k = tap * 48 + c = 96 u + 32 v + 8 g4 + e) and folded weight-norm
    w = O.pos_conv_weight(g, v)
    cg = E // G
    unpack = lambda a: a.reshape(G, Kt * cg // 96, 3, 4, cg, 8).transpose(0, 4, 1, 2, 3, 5).reshape(G, cg, Kt, cg)   # -> [g][o][tap][c]
    wp_ref = w.reshape(G, cg, cg, Kt).transpose(0, 1, 3, 2)
    assert rel_l2(unpack(wp.float().cpu().numpy()), wp_ref) < (6e-4 if dtype == "f16" else 4e-3)
    xh = round_half(x, dtype)
    out = K.posconv(_dev(xh, _tdt(dtype)), _dev(x), wp, _dev(bias), G, Kt).cpu().numpy()
    wq = unpack(wp.float().cpu().numpy()).transpose(0, 1, 3, 2).reshape(E, cg, Kt)
    conv = O.pos_conv(xh, wq, bias.astype(np.float32), G)
    assert rel_l2(out - x, conv) < 2e-5       # same rounded operands: fp32 accumulation order only
    assert out.shape == (B, T, E)
    # operand-type residual and output (the "half" residual stream mode)
    outh = K.posconv(_dev(xh, _tdt(dtype)), None, wp, _dev(bias), G, Kt, half_out=True)
    assert outh.dtype == _tdt(dtype)
    assert np.array_equal(outh.float().cpu().numpy(), round_half((xh + conv).astype(np.float32), dtype)) or \
        rel_l2(outh.float().cpu().numpy(), xh + conv) < (6e-4 if dtype == "f16" else 4e-3)


def _fbank_close(a, b, what):
    """log-mel comparison that is meaningful for tonal inputs: bins whose energy sits at the fp32 FFT
    noise floor of the frame (1e-6 of the frame's peak mel energy... the reference's own CPU/GPU test
    uses atol 2e-2) are compared in the energy domain."""
    ea, eb = np.exp(a.astype(np.float64)), np.exp(b.astype(np.float64))
    peak = eb.max(axis=-1, keepdims=True)
    err = np.abs(ea - eb) / (eb + 3e-6 * peak)
    assert err.max() < 2e-3, f"{what}: energy-domain error {err.max():.3e}"


def test_fbank_golden(built_lib, golden_dir):
    from avex_amd import kernels as K
    g = np.load(f"{golden_dir}/fbank.npz")
    plan = K.FbankPlan()
    y = plan(_dev(synth.noise_clips(2, 16000, seed=0))).cpu().numpy()
    assert y.shape == g["noise16k"].shape == (2, 98, 128)
    assert max_abs(y, g["noise16k"]) < 1e-3          # reference CPU-vs-kaldi tolerance is 1e-4 rel+abs on ~20
    _fbank_close(y, g["noise16k"], "noise16k")
    y = plan(_dev(synth.tone_clips(16000))).cpu().numpy()
    _fbank_close(y, g["tone16k"], "tone16k")
    y = plan(_dev(synth.noise_clips(2, 160000, seed=0))).cpu().numpy()
    assert y.shape == (2, 998, 128)
    assert max_abs(y[:, ::37], g["noise160k_rows37"]) < 1e-3
    assert max_abs(y.sum(-1), g["noise160k_framesum"]) < 2e-2
    imp = np.zeros((1, 16000), np.float32); imp[0, 200] = 1.0
    y = plan(_dev(imp)).cpu().numpy()
    _fbank_close(y[:, :3], g["impulse200"], "impulse")
    # DC / silence -> log(eps) everywhere
    y = plan(_dev(np.full((1, 16000), 0.25, np.float32))).cpu().numpy()
    assert np.allclose(y, math.log(1.1920929e-07), atol=1e-5)


@pytest.mark.parametrize("n_mels", [64, 256])
def test_fbank_other_mel_sizes(built_lib, golden_dir, n_mels):
    from avex_amd import kernels as K
    g = np.load(f"{golden_dir}/fbank.npz")
    plan = K.FbankPlan(n_mels=n_mels, mel_fb=K.kaldi_mel_filterbank(n_mels))
    y = plan(_dev(synth.noise_clips(1, 4000, seed=7))).cpu().numpy()
    assert y.shape == g[f"noise4k_mel{n_mels}"].shape == (1, 23, n_mels)
    assert max_abs(y, g[f"noise4k_mel{n_mels}"]) < 1e-3


def test_fbank_edge_lengths(built_lib):
    from avex_amd import kernels as K
    plan = K.FbankPlan()
    for T, frames in ((399, 0), (400, 1), (559, 1), (560, 2), (4000, 23)):
        y = plan(_dev(synth.noise_clips(1, T, seed=1)))
        assert y.shape == (1, frames, 128)
        if frames:
            ref = O.fbank(synth.noise_clips(1, T, seed=1) * np.float32(2 ** 15))
            assert max_abs(y.cpu().numpy(), ref) < 1e-3


def test_fbank_silent_frame_beside_a_loud_one(built_lib):
    """Two frames share one complex FFT; an all-zero frame packed beside a full-scale impulse must still come out at the log
    floor in every bin (the reference transforms every frame alone: beats.py:154), not at the partner's rounding noise.  Also a
    clip whose second half is zero padding, and a constant (zero after the per-frame DC removal)."""
    from avex_amd import kernels as K
    n = 16000
    x = np.zeros((3, n), np.float32)
    x[0, 12345 - 8000] = 1.0
    x[1, : n // 2] = synth.noise_clips(1, n // 2, seed=3)[0] * np.float32(5.0)
    x[2] = 0.25
    got = K.FbankPlan()(_dev(x)).cpu().numpy()
    ref = O.fbank(x * np.float32(2 ** 15))
    floor = np.float32(np.log(np.float32(1.1920929e-07)))
    silent = (ref == floor).all(axis=-1)                          # frames the reference puts at the floor everywhere
    assert silent[0].sum() > 90 and silent[1].sum() > 45 and silent[2].all()
    assert (got[silent] == floor).all()
    assert max_abs(got[~silent], ref[~silent]) < 1e-3


def test_stft_silent_frame_beside_a_loud_one(built_lib):
    """The same for the STFT frontend (frame pairs per complex FFT): zero power, not leakage, for all-zero frames."""
    from avex_amd import kernels as K
    x = np.zeros((2, 24000), np.float32)
    x[0, :9000] = synth.noise_clips(1, 9000, seed=5)[0] * np.float32(8.0)
    x[1, 15000] = 1.0
    for n_fft, hop in ((800, 160), (512, 128)):
        spec = K.MelspecPlan(n_fft=n_fft, hop_length=hop, mel=False, normalize=False)(_dev(x)).cpu().numpy()
        ts = torch.stft(torch.from_numpy(x), n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft), center=True, return_complex=True).abs().pow(2).numpy()
        zero = (ts == 0).all(axis=1)                              # [clip, frame]
        assert zero.sum() > 50
        assert (spec.transpose(0, 2, 1)[zero] == 0).all()
        assert np.abs(spec - ts).max() <= 3e-5 * ts.max()


# ------------------------------------------------------------------------------------------------------------------------
# AVES / wav2vec2 path: parity unpinned vs torchaudio (absent); checker = oracle/aves_oracle.py
# ------------------------------------------------------------------------------------------------------------------------
def test_wavconv0_matches_oracle(built_lib):
    from avex_amd import kernels as K
    from oracle import aves_oracle as AO
    sd = synth.aves_state_dict()
    x = synth.noise_clips(2, 16000, seed=41) + np.float32(0.02)
    cfg1 = dict(synth.AVES_BASE_CFG, extractor_conv_layer_config=[[512, 10, 5]])
    ref = AO.feature_extractor(x, sd, cfg1)                                     # [2, 3199, 512]
    F = ref.shape[1]
    out = K.wavconv0(_dev(x), _dev(sd["model.feature_extractor.conv_layers.0.conv.weight"].reshape(512, 10)),
                     _dev(sd["model.feature_extractor.conv_layers.0.layer_norm.weight"]),
                     _dev(sd["model.feature_extractor.conv_layers.0.layer_norm.bias"]), frames_pad=F + 9)
    got = out[:2 * (F + 9)].view(2, F + 9, 512).float().cpu().numpy()
    assert rel_l2(got[:, :F], ref) < 6e-4                                       # one rounding to f16
    assert np.all(got[:, F:] == 0)


def _wavconv0_reference(x, w, g, b):
    """Conv1d(1, 512, 10, stride 5) -> GroupNorm over time per (clip, channel), eps 1e-5 -> erf GELU, in fp64."""
    from scipy.special import erf
    F = (x.shape[1] - 10) // 5 + 1
    idx = 5 * np.arange(F)[:, None] + np.arange(10)[None, :]
    y = x.astype(np.float64)[:, idx] @ w.astype(np.float64).T
    z = (y - y.mean(axis=1, keepdims=True)) / np.sqrt(y.var(axis=1, keepdims=True) + 1e-5) * g + b
    return 0.5 * z * (1.0 + erf(z / np.sqrt(2.0)))


@pytest.mark.parametrize("T", [10, 14, 15, 5125, 5130, 10250])      # 1, 1, 2 frames; 1 024 frames (one block), 1 025; two blocks and one frame
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_wavconv0_block_edges(built_lib, T, dtype):
    from avex_amd import kernels as K
    rng = np.random.default_rng(T)
    x = (0.1 * rng.standard_normal((3, T)) + 0.01).astype(np.float32)
    w = (0.3 * rng.standard_normal((512, 10))).astype(np.float32)
    g = (1.0 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    ref = _wavconv0_reference(x, w, g, b)
    F = ref.shape[1]
    out = K.wavconv0(_dev(x), _dev(w), _dev(g), _dev(b), frames_pad=F + 5, dtype=dtype)
    got = out[:3 * (F + 5)].view(3, F + 5, 512).float().cpu().numpy()
    tol = 6e-4 if dtype == "f16" else 5e-3
    assert np.abs(got[:, :F] - ref).max() <= tol * max(1.0, np.abs(ref).max()) * 4
    if F > 1:
        assert rel_l2(got[:, :F], ref) < tol
    assert np.all(got[:, F:] == 0)


def test_wavconv0_statistics_survive_a_dc_offset_and_a_long_clip(built_lib):
    """The layer's GroupNorm statistics come from 65 second-order moments of the clip (the layer is linear in the waveform).  The hard case for
    that route: filters that cancel most of their input -- every channel's taps sum to zero -- on a clip that is almost all offset, so a
    channel's sum of squares is ~1e-7 of the moments it is computed from; 10 s of audio so that the moments run over 32 k frames."""
    from avex_amd import kernels as K
    rng = np.random.default_rng(8)
    T = 160000
    x = (0.5 + 1e-3 * rng.standard_normal((2, T))).astype(np.float32)
    x[1] *= -1.0
    w = rng.standard_normal((512, 10)).astype(np.float32)
    w -= w.mean(axis=1, keepdims=True)                                          # zero DC gain, up to fp32 rounding of the taps
    g = (1.0 + 0.1 * rng.standard_normal(512)).astype(np.float32)
    b = (0.1 * rng.standard_normal(512)).astype(np.float32)
    F = (T - 10) // 5 + 1
    idx = 5 * np.arange(F)[:, None] + np.arange(10)[None, :]
    y = x.astype(np.float64)[:, idx] @ w.astype(np.float64).T                    # [2, F, 512]
    mean, var = y.mean(axis=1, keepdims=True), y.var(axis=1, keepdims=True)
    assert np.sqrt(var).max() < 1e-2                                            # the outputs really are tiny beside the 0.5 offset
    z = (y - mean) / np.sqrt(var + 1e-5) * g + b
    from scipy.special import erf
    ref = 0.5 * z * (1.0 + erf(z / np.sqrt(2.0)))
    out = K.wavconv0(_dev(x), _dev(w), _dev(g), _dev(b), frames_pad=F + 3)
    got = out[:2 * (F + 3)].view(2, F + 3, 512).float().cpu().numpy()
    assert rel_l2(got[:, :F], ref) < 6e-4
    assert np.abs(got[:, :F] - ref).max() < 4e-3                               # f16 rounding of values up to ~4


@pytest.mark.parametrize("samples", [16000, 40000])
def test_aves_encoder_matches_oracle(built_lib, samples):
    """Conv feature extractor as strided-row GEMMs + the transformer on the shared kernels vs the NumPy restatement
    (2 transformer layers keep the CPU side short; all widths are the real ones)."""
    from avex_amd.aves_encoder import AvesEncoder, conv_frame_plan
    from oracle import aves_oracle as AO
    cfg = dict(synth.AVES_BASE_CFG, encoder_num_layers=2)
    sd = synth.aves_state_dict(cfg)
    enc = AvesEncoder(cfg, sd)
    x = synth.noise_clips(3, samples, seed=43)
    F, P = conv_frame_plan(samples, enc.convs)
    assert F[-1] == (49 if samples == 16000 else 124) and all(p >= f for p, f in zip(P, F))
    feats = enc.extract_conv_features(_dev(x)).float().cpu().numpy()
    assert rel_l2(feats, AO.feature_extractor(x, sd, cfg)) < 3e-3               # 7 layers of f16 operands
    ref, taps = AO.aves_forward(x, sd, cfg)
    r = enc.forward(_dev(x), hook_layers=[0, 1], want_features=True, want_pooled=True)
    assert r["features"].shape == ref.shape
    assert rel_l2(r["pooled"].cpu().numpy(), ref.mean(1)) < 2e-3
    assert rel_l2(r["features"].cpu().numpy(), ref) < 6e-3
    for i in (0, 1):
        name = f"model.encoder.transformer.layers.{i}.feed_forward.output_dense"
        assert rel_l2(r["hooks"][i].cpu().numpy().mean(1), taps[name].mean(1)) < 3e-3


def test_aves_full_depth_and_long_clip(built_lib):
    """All 12 transformer layers (the configuration the reference builds, aves_model.py:19-47) on 2 s clips, and a 12 s clip (599 frames:
    past the single-block attention limit) with 2 layers, vs the NumPy restatement."""
    from avex_amd.aves_encoder import AvesEncoder
    from oracle import aves_oracle as AO
    cfg = synth.AVES_BASE_CFG
    sd = synth.aves_state_dict(cfg)
    enc = AvesEncoder(cfg, sd)
    x = synth.noise_clips(2, 32000, seed=44)
    ref, taps = AO.aves_forward(x, sd, cfg)
    r = enc.forward(_dev(x), hook_layers=[0, 5, 11], want_features=True, want_pooled=True)
    assert r["features"].shape == ref.shape == (2, 99, 768)
    assert rel_l2(r["pooled"].cpu().numpy(), ref.mean(1)) < 2e-3
    assert rel_l2(r["features"].cpu().numpy(), ref) < 8e-3
    for i in (0, 5, 11):
        name = f"model.encoder.transformer.layers.{i}.feed_forward.output_dense"
        assert rel_l2(r["hooks"][i].cpu().numpy().mean(1), taps[name].mean(1)) < 3e-3
    cfg2 = dict(cfg, encoder_num_layers=2)
    sd2 = synth.aves_state_dict(cfg2)
    enc2 = AvesEncoder(cfg2, sd2)
    xl = synth.noise_clips(1, 192000, seed=45)
    refl, _ = AO.aves_forward(xl, sd2, cfg2)
    rl = enc2.forward(_dev(xl), want_features=True, want_pooled=True)
    assert rl["features"].shape == refl.shape == (1, 599, 768)
    assert rel_l2(rl["pooled"].cpu().numpy(), refl.mean(1)) < 2e-3


def test_aves_at_bench_size_against_the_oracle_vectors(built_lib):
    """AVES (wav2vec2-base, 12 layers) at the size scripts/aves_bench.py times -- 128 clips x 10 s, 499 frames each -- with the two clips of
    tests/golden/family_small.npz (oracle/aves_oracle.py on the synthetic checkpoint; UNPINNED: torchaudio is absent from the reference tree) at rows
    0 and 127: token mean on the operand-type residual stream, an un-averaged frame on the fp32 stream (residual="auto"); and clips are independent."""
    import os
    from avex_amd.aves_encoder import AvesEncoder
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "family_small.npz"))
    cfg = synth.AVES_BASE_CFG
    enc = AvesEncoder(cfg, synth.aves_state_dict(cfg))
    n = int(gold["aves.samples"][0])
    wav = torch.from_numpy(synth.noise_clips(128, n, seed=4)).cuda()
    g = torch.from_numpy(synth.noise_clips(2, n, seed=int(gold["aves.seed"][0]))).cuda()
    wav[0] = g[0]; wav[127] = g[1]
    pooled = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    assert pooled.shape == (128, 768) and torch.isfinite(pooled).all()
    assert rel_l2(torch.stack([pooled[0], pooled[127]]).cpu().numpy(), gold["aves.pooled"]) < 1e-3
    two = enc.forward(g, want_features=False, want_pooled=True)["pooled"]
    assert rel_l2(two.cpu().numpy(), torch.stack([pooled[0], pooled[127]]).cpu().numpy()) < 5e-4      # (small batches take other kernels: other roundings of the same arithmetic)
    feats = enc.forward(g, want_features=True)["features"]
    assert rel_l2(feats[:, 0].cpu().numpy(), gold["aves.frame0"]) < 2e-3


# ------------------------------------------------------------------------------------------------------------------------
# Spectrogram / mel frontend of the reference's AudioProcessor
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("samples", [16000, 160000])
def test_melspec_matches_oracle_effnet_config(built_lib, samples):
    """EfficientNet settings (n_fft 800, hop 160, periodic Hann, 128 HTK mels, centre, log + min-max) through the C ABI vs
    the NumPy restatement whose STFT stage is pinned against torch.stft; and the torch.stft power directly for the plain
    spectrogram."""
    from avex_amd import kernels as K
    x = synth.noise_clips(2, samples, seed=61) * np.float32(3.0)
    xd = _dev(x)
    plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
    y = plan(xd)
    ref = O.audio_processor(x, n_fft=800, hop=160)
    assert y.shape == ref.shape == (2, 128, 1 + samples // 160)
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-4            # values in [0, 1] after min-max
    raw = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=False)(xd).cpu().numpy()
    ref_raw = O.audio_processor(x, n_fft=800, hop=160, normalize=False)
    assert rel_l2(raw, ref_raw) < 2e-5
    spec = K.MelspecPlan(n_fft=800, hop_length=160, mel=False, normalize=False)(xd).cpu().numpy()
    ts = torch.stft(torch.from_numpy(x), n_fft=800, hop_length=160, win_length=800, window=torch.hann_window(800), center=True,
                    return_complex=True).abs().pow(2).numpy()
    assert spec.shape == ts.shape == (2, 401, 1 + samples // 160)
    assert np.abs(spec - ts).max() <= 3e-5 * ts.max()


def test_melspec_other_configs_and_processor_mirror(built_lib):
    from avex_amd import kernels as K
    from avex_amd.base_model import AudioProcessor
    from avex_amd.configs import AudioConfig
    x = synth.noise_clips(3, 20000, seed=62)
    y = K.MelspecPlan(n_fft=512, hop_length=128, win_length=400, window="hamming", n_mels=64, center=False, normalize=True)(_dev(x)).cpu().numpy()
    ref = O.audio_processor(x, n_fft=512, hop=128, win_length=400, window="hamming", n_mels=64, center=False)
    assert y.shape == ref.shape and np.abs(y - ref).max() < 2e-4
    proc = AudioProcessor(AudioConfig(sample_rate=16000, n_fft=800, hop_length=160, win_length=800, window="hann", n_mels=128,
                                      representation="mel_spectrogram", normalize=True, center=True))
    out = proc(torch.from_numpy(x))                                  # CPU in -> CPU out, like the reference
    assert not out.is_cuda and out.shape == (3, 128, 126)
    assert np.abs(out.numpy() - O.audio_processor(x, n_fft=800, hop=160)).max() < 2e-4
    from avex_amd._capi import AvexHipError
    with pytest.raises(AvexHipError):
        K.MelspecPlan(n_fft=4096, hop_length=1024)                    # beyond the built sizes: refuses loudly
    with pytest.raises(AvexHipError):
        K.MelspecPlan(n_fft=1078, hop_length=256)                     # 2 * 7^2 * 11: no FFT radix, too long for the dense product


@pytest.mark.parametrize("n_fft,hop,win,window,center", [(2048, 512, 2048, "hann", True), (2048, 512, 1200, "hamming", True), (1024, 256, 1024, "hann", False),
                                                          (400, 160, 400, "hann", True), (96, 24, 96, "hann", True), (1500, 300, 1500, "hann", True),
                                                          (800, 160, 800, "hann", True)])
def test_stft_fft_path_matches_torch_stft(built_lib, n_fft, hop, win, window, center):
    """The mixed-radix FFT frontend (n_fft = 2^a 3^b 5^c up to 2048, incl. the AudioConfig default 2048 / hop 512, configs.py:178) against
    torch.stft itself: plain power spectrogram, then the whole AudioProcessor chain against the oracle."""
    from avex_amd import kernels as K
    x = synth.noise_clips(2, 24000, seed=63) * np.float32(2.0)
    xd = _dev(x)
    spec = K.MelspecPlan(n_fft=n_fft, hop_length=hop, win_length=win, window=window, mel=False, center=center, normalize=False)(xd).cpu().numpy()
    wt = torch.hann_window(win) if window == "hann" else torch.hamming_window(win)
    ts = torch.stft(torch.from_numpy(x), n_fft=n_fft, hop_length=hop, win_length=win, window=wt, center=center, return_complex=True).abs().pow(2).numpy()
    assert spec.shape == ts.shape
    assert np.abs(spec - ts).max() <= 3e-5 * ts.max()
    y = K.MelspecPlan(n_fft=n_fft, hop_length=hop, win_length=win, window=window, n_mels=64, center=center, normalize=True)(xd).cpu().numpy()
    ref = O.audio_processor(x, n_fft=n_fft, hop=hop, win_length=win, window=window, n_mels=64, center=center)
    assert y.shape == ref.shape and np.abs(y - ref).max() < 2e-4


def test_stft_in_place_passes_give_the_bits_of_the_ping_pong_passes(built_lib, tmp_path):
    """n_fft 800 (the EfficientNet frontend) runs its compiled-in Stockham passes IN PLACE in one LDS buffer per wave; AVEX_AMD_STFT_GENERIC=1
    selects the run-time passes with their ping-pong pair.  Same operations in the same order: the log-mel outputs must be bit-identical.
    (The switch is read once per process: two fresh processes.)"""
    import subprocess, sys
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from avex_amd import synth, kernels as K\n"
        "x = torch.from_numpy(synth.noise_clips(3, 48000, seed=65)).cuda()\n"
        "y = K.MelspecPlan(n_fft=800, hop_length=320, n_mels=128, normalize=True)(x)\n"
        "p = K.MelspecPlan(n_fft=800, hop_length=160, mel=False, normalize=False)(x)\n"
        "np.savez(sys.argv[1], y=y.cpu().numpy(), p=p.cpu().numpy())\n")
    outs = []
    for generic in ("0", "1"):
        o = str(tmp_path / f"g{generic}.npz")
        r = subprocess.run([sys.executable, "-c", code, o], capture_output=True, text=True, timeout=600, env=dict(os.environ, AVEX_AMD_STFT_GENERIC=generic))
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append(np.load(o))
    for k in ("y", "p"):
        assert np.isfinite(outs[0][k]).all() and np.array_equal(outs[0][k], outs[1][k]), k


def test_stft_fft_and_dense_paths_agree(built_lib, monkeypatch):
    """n_fft 800 both ways: the FFT (default) and the dense fp32-MFMA product (AVEX_AMD_MELSPEC_DENSE=1, the fallback for lengths
    with prime factors above 5), and a length only the dense path takes (770 = 2 * 5 * 7 * 11)."""
    from avex_amd import kernels as K
    x = synth.noise_clips(2, 32000, seed=64)
    xd = _dev(x)
    a = K.MelspecPlan(n_fft=800, hop_length=160, mel=False, normalize=False)(xd).cpu().numpy()
    monkeypatch.setenv("AVEX_AMD_MELSPEC_DENSE", "1")
    b = K.MelspecPlan(n_fft=800, hop_length=160, mel=False, normalize=False)(xd).cpu().numpy()
    monkeypatch.delenv("AVEX_AMD_MELSPEC_DENSE")
    assert np.abs(a - b).max() <= 3e-5 * b.max()
    c = K.MelspecPlan(n_fft=770, hop_length=160, mel=False, normalize=False)(xd).cpu().numpy()
    ts = torch.stft(torch.from_numpy(x), n_fft=770, hop_length=160, win_length=770, window=torch.hann_window(770), center=True, return_complex=True).abs().pow(2).numpy()
    assert np.abs(c - ts).max() <= 3e-5 * ts.max()


# ------------------------------------------------------------------------------------------------------------------------
# EfficientNet-B0: parity unpinned vs torchvision (absent); checker = oracle/effnet_oracle.py
# ------------------------------------------------------------------------------------------------------------------------
def test_effnet_blocks_match_oracle(built_lib):
    """Stem, depthwise (3x3 s1, 5x5 s2) + squeeze pool, squeeze-excitation against NumPy on the same folded parameters."""
    from avex_amd import kernels as K
    from oracle import effnet_oracle as EO
    B, H, W, C, Cp = 2, 20, 37, 40, 128
    img = synth.normal("eimg", (B, H, W), 1.0)
    w = synth.normal("estem", (C, 3, 3), 0.4); bias = synth.normal("estemb", (C,), 0.2)
    ws = np.zeros((9, Cp), np.float32); ws[:, :C] = w.reshape(C, 9).T
    bs = np.zeros((Cp,), np.float32); bs[:C] = bias
    y = K.effnet_stem(_dev(img), _dev(ws), _dev(bs)).float().cpu().numpy()
    ref = EO.silu(EO.conv2d(img[:, None], w[:, None], 2, 1) + bias[None, :, None, None]).transpose(0, 2, 3, 1)
    assert y.shape == (B, 10, 19, Cp) and rel_l2(y[..., :C], ref) < 6e-4 and np.all(y[..., C:] == 0)
    x = round_half(np.pad(ref, ((0, 0), (0, 0), (0, 0), (0, Cp - C))), "f16")
    for k, s in ((3, 1), (5, 2), (3, 2), (5, 1)):
        wd = synth.normal(f"edw{k}{s}", (C, k, k), 0.3); bd = synth.normal("edwb", (C,), 0.1)
        wdp = np.zeros((k * k, Cp), np.float32); wdp[:, :C] = wd.reshape(C, k * k).T
        bdp = np.zeros((Cp,), np.float32); bdp[:C] = bd
        out, pool = K.effnet_dwconv(_dev(x, torch.float16), _dev(wdp), _dev(bdp), k, s)
        r = EO.silu(EO.conv2d(x[..., :C].transpose(0, 3, 1, 2), wd[:, None], s, (k - 1) // 2, groups=C) + bd[None, :, None, None]).transpose(0, 2, 3, 1)
        o = out.float().cpu().numpy()
        assert o.shape[:3] == r.shape[:3] and rel_l2(o[..., :C], r) < 6e-4 and np.all(o[..., C:] == 0)
        assert np.allclose(pool.cpu().numpy()[:, :C], o[..., :C].sum((1, 2)), rtol=1e-4, atol=1e-2)
        for _ in range(3):                                  # ordered partial sums, not atomics: the squeeze repeats bit for bit
            assert torch.equal(K.effnet_dwconv(_dev(x, torch.float16), _dev(wdp), _dev(bdp), k, s)[1], pool)
        Cs = 10
        w1 = synth.normal("ese1", (Cs, C), 0.3); b1 = synth.normal("ese1b", (Cs,), 0.1); w2 = synth.normal("ese2", (C, Cs), 0.5); b2 = synth.normal("ese2b", (C,), 0.3)
        xs = out.clone()
        sc = K.effnet_se(xs, pool, C, _dev(w1), _dev(b1), _dev(w2), _dev(b2)).cpu().numpy()
        m = o[..., :C].mean((1, 2))
        e = 1.0 / (1.0 + np.exp(-(EO.silu(m @ w1.T + b1) @ w2.T + b2)))
        assert np.allclose(sc[:, :C], e, rtol=2e-4, atol=2e-5) and np.all(sc[:, C:] == 0)
        assert rel_l2(xs.float().cpu().numpy()[..., :C], o[..., :C] * e[:, None, None, :]) < 6e-4


def test_effnet_b1_matches_oracle(built_lib):
    """EfficientNet-B1 (efficientnet.py:64-66: torchvision efficientnet_b1 = B0's widths with ceil(1.1 n) blocks per stage) through
    the same kernels, against the NumPy restatement."""
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    from oracle import effnet_oracle as EO
    sd = synth.effnet_b0_state_dict(stages=synth.EFFNET_B1_STAGES)
    enc = EfficientNetB0Encoder(sd, stages=synth.EFFNET_B1_STAGES)
    mel = np.abs(synth.normal("emel1", (2, 64, 101), 0.5)).astype(np.float32)
    ref, taps = EO.effnet_features(mel, sd, synth.EFFNET_B1_STAGES)
    names = enc.tap_names()
    assert names == list(taps.keys()) and len(names) == 23
    r = enc.forward(_dev(mel), hook_layers=[names[0], names[-1]], want_features=True, want_pooled=True)
    assert r["features"].shape == ref.shape == (2, 1280, 2, 4)
    assert rel_l2(r["features"].cpu().numpy(), ref) < 1.5e-3
    assert rel_l2(r["pooled"].cpu().numpy(), ref.mean((2, 3))) < 1e-3


def test_effnet_b0_matches_oracle(built_lib):
    """The whole features stack (all 16 MBConv blocks, real widths) on a small mel image vs the NumPy restatement."""
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    from oracle import effnet_oracle as EO
    sd = synth.effnet_b0_state_dict()
    enc = EfficientNetB0Encoder(sd)
    mel = np.abs(synth.normal("emel", (2, 64, 101), 0.5)).astype(np.float32)
    ref, taps = EO.effnet_features(mel, sd, synth.EFFNET_B0_STAGES)
    names = enc.tap_names()
    assert names == list(taps.keys()) and len(names) == 17
    r = enc.forward(_dev(mel), hook_layers=[names[0], names[3], names[-1]], want_features=True, want_pooled=True)
    assert r["features"].shape == ref.shape == (2, 1280, 2, 4)
    # measured (tests/tools/effnet_tap_errors.py, f16): every tap 4e-4 .. 6e-4, features 3.5e-4, pooled 2.2e-4 -- the rounding of one layer's
    # operands; eval-mode BatchNorm renormalises every layer, so nothing accumulates over the ~50 layers (bf16: 4e-3 / 2.8e-3 / 1.8e-3)
    assert rel_l2(r["features"].cpu().numpy(), ref) < 1.2e-3
    assert rel_l2(r["pooled"].cpu().numpy(), ref.mean((2, 3))) < 8e-4
    assert rel_l2(r["hooks"][names[0]].cpu().numpy(), taps[names[0]]) < 2e-5     # stem tap: fp32
    assert rel_l2(r["hooks"][names[3]].cpu().numpy(), taps[names[3]]) < 1.5e-3
    assert rel_l2(r["hooks"][names[-1]].cpu().numpy(), taps[names[-1]]) < 1.5e-3
    allr = enc.forward(_dev(mel), hook_layers=names, want_features=False)
    for n in names[1:]:
        assert rel_l2(allr["hooks"][n].cpu().numpy(), taps[n]) < 1.5e-3, n
    again = enc.forward(_dev(mel), hook_layers=[names[-1]], want_features=True, want_pooled=True)      # no float atomics anywhere: bit for bit
    assert torch.equal(again["features"], r["features"]) and torch.equal(again["pooled"], r["pooled"])
    assert torch.equal(again["hooks"][names[-1]], r["hooks"][names[-1]])


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("stages", ["b0", "b1"])
def test_effnet_fused_block_front_matches_the_unfused_kernels(built_lib, monkeypatch, dtype, stages):
    """avx::mbconv_front (expansion + depthwise convolution + squeeze sums in one kernel, effnet.hip) and avx::dwconv_lds_parts (the
    depthwise convolution through LDS) against the expansion GEMM and dwconv_kernel they replace (AVEX_AMD_MBCONV=0, AVEX_AMD_DW_LDS=0), on an image whose sizes are not multiples of any tile (ragged tiles on both axes, halos
    that leave the image on all four sides).  Both forms round at the same places and add their taps in the same order; what differs is
    the order in which the squeeze partial sums are added (one row per workgroup, different workgroups), i.e. fp32 rounding of the
    squeeze-excitation scale, which flips an occasional operand rounding downstream."""
    from avex_amd.effnet_encoder import EfficientNetB0Encoder
    st = synth.EFFNET_B0_STAGES if stages == "b0" else synth.EFFNET_B1_STAGES
    sd = synth.effnet_b0_state_dict(stages=st) if stages == "b1" else synth.effnet_b0_state_dict()
    kw = {"stages": st} if stages == "b1" else {}
    mel = _dev(np.abs(synth.normal("emelf", (3, 70, 133), 0.5)).astype(np.float32))
    enc = EfficientNetB0Encoder(sd, operand_dtype=dtype, **kw)
    names = enc.tap_names()
    monkeypatch.setenv("AVEX_AMD_DW_LDS", "2")                    # the LDS depthwise kernel for every block it can take, small maps and stride 2 too
    fused = enc.forward(mel, hook_layers=names, want_features=True, want_pooled=True)
    monkeypatch.setenv("AVEX_AMD_MBCONV", "0")
    monkeypatch.setenv("AVEX_AMD_DW_LDS", "0")                    # ... and the register-tile depthwise kernel instead of the LDS one for the other blocks
    plain = enc.forward(mel, hook_layers=names, want_features=True, want_pooled=True)
    monkeypatch.delenv("AVEX_AMD_MBCONV")
    monkeypatch.delenv("AVEX_AMD_DW_LDS")
    tol = 2e-4 if dtype == "f16" else 1.5e-3
    assert torch.equal(fused["hooks"][names[0]], plain["hooks"][names[0]])          # the stem is not touched
    for n in names[1:]:
        assert rel_l2(fused["hooks"][n].cpu().numpy(), plain["hooks"][n].cpu().numpy()) < tol, n
    assert rel_l2(fused["pooled"].cpu().numpy(), plain["pooled"].cpu().numpy()) < tol
    default = enc.forward(mel, hook_layers=names, want_features=True, want_pooled=True)      # the shipped choice per block
    assert rel_l2(default["pooled"].cpu().numpy(), plain["pooled"].cpu().numpy()) < tol
    again = enc.forward(mel, hook_layers=names, want_features=True, want_pooled=True)
    assert torch.equal(again["features"], default["features"]) and torch.equal(again["pooled"], default["pooled"])


def test_effnet_random_layouts_fuzz(built_lib):
    """tests/tools/fuzz_effnet.py: ten random stage layouts / image sizes through the handle -- shipped kernels vs the unfused forms vs
    the oracle, and repeatability."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_effnet.py"), "10", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
