import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from avex_amd import kernels as K, synth
cfg = synth.BEATS_BASE_CFG; sd = synth.beats_state_dict(cfg, seed=0)
g = np.load("tests/golden/base_api.npz")["b4.pooled"]
x = torch.from_numpy(synth.noise_clips(4, 160000, seed=0)).cuda()
for B in (4, 32):
    xx = torch.cat([x] + [x[:1]] * (B - 4)) if B > 4 else x
    e = K.BeatsEncoder(cfg, sd, operand_dtype="f16")
    p = e.forward(xx, want_features=False, want_pooled=True)["pooled"][:4].cpu().numpy()
    print(os.environ.get("TAG"), "B", B, "rel", np.linalg.norm(p - g) / np.linalg.norm(g), flush=True)
    e.close()
