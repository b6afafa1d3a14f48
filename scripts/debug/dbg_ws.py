import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
os.environ["AVEX_AMD_STREAMS"] = "1"; enc1 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32", max_chunk_clips=32)
os.environ["AVEX_AMD_STREAMS"] = "2"; enc2 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32")
B = 64
A = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
M = 32 * 496
sizes = [("patches", M * 256 * 2), ("f0", M * 512 * 4), ("h0", M * 512 * 2)]
def grab(enc, lane_off):
    out = {}; off = lane_off
    for n, sz in sizes:
        out[n] = enc._ws[off:off + sz].clone(); off += al(sz)
    return out
# enc1 with chunk 32: sequential chunks reuse ONE workspace; after forward it holds chunk 1 (clips 32..63) = lane 1's data
r1 = enc1.forward(A, want_features=False, want_pooled=True); torch.cuda.synchronize()
w1 = grab(enc1, 0)
r2 = enc2.forward(A, want_features=False, want_pooled=True); torch.cuda.synchronize()
total = enc2._ws.numel() // 2
w2 = grab(enc2, total)
for n, sz in sizes:
    a, b = w1[n], w2[n]
    neq = (a != b).view(32, -1)
    print(n, "clips with diffs:", neq.any(dim=1).nonzero().flatten().tolist(), "count", int(neq.sum()))
d = (r1["pooled"] - r2["pooled"]).abs().max(dim=1)[0]
print("pooled bad:", (d > 0).nonzero().flatten().tolist())
a = w1["patches"].view(torch.float16).view(32, 496, 256).float(); b = w2["patches"].view(torch.float16).view(32, 496, 256).float()
for c in range(6):
    neq = (a[c] != b[c])
    toks = neq.any(dim=1).nonzero().flatten()
    # token = tp*8 + fq ; elem = (frame%16)*16 + mel%16
    tp = (toks // 8).unique().tolist()
    el = neq.any(dim=0).nonzero().flatten()
    fr16 = (el // 16).unique().tolist(); mel16 = (el % 16).unique().tolist()
    fq = (toks % 8).unique().tolist()
    print("clip", c, "n", int(neq.sum()), "time-patches", tp[:40], "fq", fq, "frame%16", fr16, "mel%16", mel16, "maxdiff", float((a[c]-b[c]).abs().max()))
# per-frame pattern for clip 0: reconstruct [frame, mel]
def to_fm(x):  # x [496,256] -> [992,128]
    return x.view(62, 8, 16, 16).permute(0, 2, 1, 3).reshape(992, 128)
fa, fb_ = to_fm(a[0]), to_fm(b[0])
neq = fa != fb_
cnt = neq.sum(dim=1)
bad = cnt.nonzero().flatten().tolist()
print("bad frames (first 60):", bad[:60])
print("wrong mel count per bad frame:", cnt[bad[:40]].tolist())
f = bad[0]
print("frame", f, "wrong mels:", neq[f].nonzero().flatten().tolist())
print("ref :", fa[f][neq[f]][:8].tolist()); print("got :", fb_[f][neq[f]][:8].tolist())
# does 'got' equal the reference of a neighbouring frame?
for df in (-2, -1, 1, 2):
    if 0 <= f + df < 992: print("  match with ref frame", f + df, ":", int((fb_[f][neq[f]] == fa[f + df][neq[f]]).sum()), "of", int(neq[f].sum()))
