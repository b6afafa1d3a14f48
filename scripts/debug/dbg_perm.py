import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
for res in ("f32", "half"):
    enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype="f16", residual=res)
    B = 64
    wav = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
    p = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    pp = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    wp = wav[perm]
    if os.environ.get("DBG_SYNC"): torch.cuda.synchronize()
    p2 = enc.forward(wp, want_features=False, want_pooled=True)["pooled"]
    d = (p2 - p[perm]).abs().max(dim=1)[0]
    print(os.environ.get("AVEX_AMD_STREAMS"), res, "repeat equal:", torch.equal(p, pp), "perm max diff:", float(d.max()), "bad clips:", int((d > 0).sum()), (d > 0).nonzero().flatten().tolist()[:20])
    enc.close()
