import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype="f16", residual="f32")
B = 64
wav = torch.from_numpy(synth.noise_clips(B, 160000, seed=0)).cuda()
r = enc.forward(wav, hook_layers=range(13), want_features=True, want_pooled=True)
torch.cuda.synchronize()
out = {f"h{i}": r["hooks"][i].mean(1).cpu().numpy() for i in range(13)}
out["pooled"] = r["pooled"].cpu().numpy()
out["h0_full32"] = r["hooks"][0][32].cpu().numpy()
np.savez(f"/tmp/taps_{os.environ.get('AVEX_AMD_STREAMS','1')}.npz", **out)
if os.path.exists("/tmp/taps_1.npz") and os.path.exists("/tmp/taps_2.npz"):
    a, b = np.load("/tmp/taps_1.npz"), np.load("/tmp/taps_2.npz")
    for k in a.files:
        if k == "h0_full32":
            d = np.abs(a[k] - b[k]).max(axis=1); print(k, "bad tokens:", np.nonzero(d > 0)[0][:20], d.max()); continue
        d = np.abs(a[k] - b[k]).max(axis=1)
        print(k, "bad clips:", np.nonzero(d > 0)[0].tolist()[:12], float(d.max()))
