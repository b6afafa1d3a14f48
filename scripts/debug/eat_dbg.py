import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from avex_amd import synth
from avex_amd.eat_encoder import EatEncoder
rel=lambda a,b: float(np.linalg.norm(a-b)/np.linalg.norm(b))
enc = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype="f16")
wav = torch.from_numpy(synth.noise_clips(64, 80000, seed=16)).cuda()
rows=[0,30,63]
full = enc.forward(wav, want_features=True, pooling="mean")
small = enc.forward(wav[rows].contiguous(), want_features=True, pooling="mean")
print("pooled B=64(fused? no: features wanted) vs B=3", rel(small["pooled"].cpu().numpy(), full["pooled"][rows].cpu().numpy()))
print("features", rel(small["features"].cpu().numpy(), full["features"][rows].cpu().numpy()), torch.equal(small["features"], full["features"][rows]))
fp = enc.forward(wav, want_features=False, pooling="mean")["pooled"]
print("fused pool vs mean of features", rel(fp.cpu().numpy(), full["pooled"].cpu().numpy()))
sp = enc.forward(wav[rows].contiguous(), want_features=False, pooling="mean")["pooled"]
print("small no-features vs full fused", rel(sp.cpu().numpy(), fp[rows].cpu().numpy()))
for i in (0,5,11):
    a = enc.forward(wav, hook_layers=[i], want_features=False)["hooks"][i][rows]
    b = enc.forward(wav[rows].contiguous(), hook_layers=[i], want_features=False)["hooks"][i]
    print("hook", i, rel(b.cpu().numpy(), a.cpu().numpy()), torch.equal(a,b))
os.environ["AVEX_AMD_LN_FOLD"]="0"
enc2 = EatEncoder(synth.EAT_BASE_CFG, synth.eat_state_dict(), operand_dtype="f16")
f2 = enc2.forward(wav, want_features=True)["features"]; s2 = enc2.forward(wav[rows].contiguous(), want_features=True)["features"]
print("nofold features equal", torch.equal(s2, f2[rows]), rel(s2.cpu().numpy(), f2[rows].cpu().numpy()))
