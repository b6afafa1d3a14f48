import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from avex_amd import synth, kernels as K
from oracle import beats_oracle as O
from _util import rel_l2
cfg = synth.BEATS_BASE_CFG
sd = synth.beats_state_dict(cfg, seed=0)
n = 32000
x = np.zeros((1, n), np.float32); x[0, 12345] = 1.0
f, taps = O.beats_forward(x, sd, cfg)
names = O.layer_names(cfg)
enc = K.BeatsEncoder(cfg, sd, operand_dtype="f16", max_chunk_clips=3, residual=os.environ.get("RES", "f32"))
r = enc.forward(torch.from_numpy(x).cuda(), hook_layers=list(range(13)), want_features=True, want_pooled=True)
fb_o = O.beats_preprocess(x, cfg)
plan = K.FbankPlan()
fb_g = plan(torch.from_numpy(x).cuda() ).cpu().numpy() if False else None
for i, nm in enumerate(names):
    g = r["hooks"][i].cpu().numpy()[0]; w = taps[nm][0]
    per_tok = np.linalg.norm(g - w, axis=1) / np.linalg.norm(w, axis=1)
    print(f"{i:2d} {nm:45s} rel {rel_l2(g, w):.2e}  worst tokens {np.argsort(per_tok)[-4:][::-1]} {np.sort(per_tok)[-4:][::-1].round(4)}  median {np.median(per_tok):.2e}")
g = r["features"].cpu().numpy()[0]; w = f[0]
per_tok = np.linalg.norm(g - w, axis=1) / np.linalg.norm(w, axis=1)
print("features rel", rel_l2(g, w), "worst", np.argsort(per_tok)[-6:][::-1], np.sort(per_tok)[-6:][::-1].round(4), "median", np.median(per_tok))
