"""Characterise the fbank-under-concurrency corruption: which blocks / frames / bins are wrong and what the wrong values look like."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K, _capi
L = _capi.lib()
B = 32
wav = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
plan = K.FbankPlan(norm_mean=15.41663, norm_div=13.11164)
ref = plan(wav).clone(); torch.cuda.synchronize()
frames = ref.shape[1]
M = 32 * 496
x = torch.randn(M, 768, device="cuda").half(); w = (torch.randn(2304, 768, device="cuda") * 0.05).half()
x32 = torch.randn(M, 768, device="cuda"); lw = torch.ones(768, device="cuda"); lb = torch.zeros(768, device="cuda")
big = torch.randn(64 * 1024 * 1024, device="cuda")
aggr = {
    "gemm_v3(nt,LDS-DMA)": lambda: K.gemm(x, w, out_f32=False, out_half=True, variant=3),
    "gemm_v1(nt,regs)": lambda: K.gemm(x, w, out_f32=False, out_half=True, variant=1),
    "gemm_default": lambda: K.gemm(x, w, out_f32=False, out_half=True),
    "layernorm": lambda: K.layernorm(x32, lw, lb),
    "torch_copy": lambda: big.clone(),
    "torch_matmul": lambda: x @ w.t(),
    "nothing": lambda: None,
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NR = 6
for name, fn in aggr.items():
    outs = [torch.full((B, frames, 128), float("nan"), device="cuda") for _ in range(NR)]
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(40): fn()
    with torch.cuda.stream(sb):
        for o in outs:
            _capi.check(L.avexhip_fbank_forward(plan._h, wav.data_ptr(), B, 160000, 160000, o.data_ptr(), sb.cuda_stream), "fbank")
    torch.cuda.synchronize()
    tot = 0
    for r, o in enumerate(outs):
        bad = (o != ref) | torch.isnan(o)
        nb = int(bad.sum())
        tot += nb
        if nb == 0:
            continue
        idx = bad.nonzero().cpu().numpy()
        clips = np.unique(idx[:, 0]); frs = np.unique(idx[:, 1])
        nan_ct = int(torch.isnan(o).sum())
        # frame-granularity: are whole frames wrong?  pairs (wave = 2 frames)?  blocks (8 frames)?
        fb = bad.any(dim=2).cpu().numpy()          # [B, frames]
        per_frame_bins = bad.sum(dim=2).cpu().numpy()[fb]
        blocks = np.unique((np.argwhere(fb)[:, 0] * 125 + np.argwhere(fb)[:, 1] // 8))
        d = (o - ref)[bad]
        print(f"  run {r}: wrong={nb} nan={nan_ct} clips={clips[:8]}..({len(clips)}) frames[{frs.min()}..{frs.max()}] bad_frames={int(fb.sum())} "
              f"bins/frame min/med/max={per_frame_bins.min()}/{int(np.median(per_frame_bins))}/{per_frame_bins.max()} blocks={len(blocks)} first_blocks={blocks[:6]} "
              f"|diff| med={float(d.abs().median()):.3g} max={float(d.abs().max()):.3g}")
        # does a wrong frame equal some other frame of the reference (stale / cross-wave data)?
        bi, fi = np.argwhere(fb)[0]
        row = o[bi, fi]
        dist = (ref[bi] - row[None, :]).abs().max(dim=1)[0]
        j = int(dist.argmin())
        print(f"          first bad frame clip {bi} frame {fi}: nearest ref frame {j} (maxdiff {float(dist[j]):.3g}); row[:6]={row[:6].cpu().numpy()} ref[:6]={ref[bi, fi, :6].cpu().numpy()}")
    print(f"fbank beside {name:22s}: total wrong elements over {NR} runs = {tot}")
