import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
os.environ["AVEX_AMD_STREAMS"] = "1"; enc1 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32")
os.environ["AVEX_AMD_STREAMS"] = "2"; enc2 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32")
B = 64
A = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
torch.cuda.synchronize()
cfgs = {"pooled": dict(want_features=False, want_pooled=True), "features": dict(want_features=True, want_pooled=False),
        "feat+pooled": dict(want_features=True, want_pooled=True), "hooks_pooled": dict(want_features=False, want_pooled=True, hook_layers=[0, 6, 12], hook_pooled=True),
        "hooks_full": dict(want_features=False, want_pooled=True, hook_layers=[0, 6, 12])}
def summarize(r):
    out = {}
    if r["pooled"] is not None: out["pooled"] = r["pooled"].clone()
    if r["features"] is not None: out["features"] = r["features"].mean(1)
    for i, h in r["hooks"].items(): out[f"h{i}"] = h.clone() if h.dim() == 2 else h.mean(1)
    return out
for name, kw in cfgs.items():
    r1 = summarize(enc1.forward(A, **kw))
    torch.cuda.synchronize()
    for rep in range(2):
        r2 = summarize(enc2.forward(A, **kw))
        torch.cuda.synchronize()
        msg = []
        for k in r1:
            d = (r1[k] - r2[k]).abs().max(dim=1)[0]
            msg.append(f"{k}: bad {(d > 0).nonzero().flatten().tolist()}")
        print(name, "rep", rep, " | ".join(msg))
