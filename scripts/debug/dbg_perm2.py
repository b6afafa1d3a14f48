import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
os.environ["AVEX_AMD_STREAMS"] = "1"; enc1 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32")
os.environ["AVEX_AMD_STREAMS"] = "2"; enc2 = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, residual="f32")
B = 64
A = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
inputs = {"A": A, "A.clone": A.clone(), "A[perm]": A[perm], "A[perm].clone": A[perm].clone(), "A[arange]": A[torch.arange(B).cuda()]}
torch.cuda.synchronize()
for name, X in inputs.items():
    r1 = enc1.forward(X, want_features=False, want_pooled=True)["pooled"].clone()
    for rep in range(2):
        r2 = enc2.forward(X, want_features=False, want_pooled=True)["pooled"].clone()
        d = (r1 - r2).abs().max(dim=1)[0]
        print(name, "rep", rep, "ptr%4096:", X.data_ptr() % 4096, "bad:", (d > 0).nonzero().flatten().tolist(), float(d.max()))
