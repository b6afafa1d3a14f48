import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # tests/tools/: may import the oracle
sys.path.insert(0, ROOT)
import numpy as np, torch
from avex_amd import synth
from avex_amd.effnet_encoder import EfficientNetB0Encoder
from oracle import effnet_oracle as EO
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
stages = [(1, 5, 2, 32, 24, 1), (4, 3, 1, 24, 8, 2)]
rng = np.random.default_rng(11)
# replay the fuzzer's draws up to case 1
sd = synth.effnet_b0_state_dict(seed=1, stages=stages, head=1280)
for H, W in ((54, 126), (56, 128), (54, 128), (56, 126)):
    mel = np.abs(rng.standard_normal((1, H, W))).astype(np.float32) * np.float32(0.5)
    x = torch.from_numpy(mel).cuda()
    enc = EfficientNetB0Encoder(sd, operand_dtype="f16", stages=stages)
    names = enc.tap_names()
    ref, taps = EO.effnet_features(mel, sd, stages)
    out = {}
    for mode, env in (("lds2", {"AVEX_AMD_DW_LDS": "2"}), ("default", {}), ("plain", {"AVEX_AMD_DW_LDS": "0", "AVEX_AMD_MBCONV": "0"}), ("lds2_nofuse", {"AVEX_AMD_DW_LDS": "2", "AVEX_AMD_MBCONV": "0"})):
        for k in ("AVEX_AMD_DW_LDS", "AVEX_AMD_MBCONV"): os.environ.pop(k, None)
        os.environ.update(env)
        out[mode] = enc.forward(x, hook_layers=names, want_features=True, want_pooled=True)
    for k in ("AVEX_AMD_DW_LDS", "AVEX_AMD_MBCONV"): os.environ.pop(k, None)
    print(f"{H}x{W}:", {m: [round(rel(out[m]['hooks'][n].cpu().numpy(), taps[n]), 5) for n in names] for m in out})
a = out["default"]["hooks"][names[-1]].cpu().numpy(); b = out["plain"]["hooks"][names[-1]].cpu().numpy()
d = np.abs(a - b)
print("head tap shape", a.shape, "max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
bad = np.argwhere(d > 1e-3 * np.abs(b).max())
print("bad elements", len(bad), "distinct channels", len(set(bad[:,1].tolist())), "rows", sorted(set(bad[:,2].tolist())), "cols", sorted(set(bad[:,3].tolist())))
a2 = out["default"]["hooks"][names[-2]].cpu().numpy(); b2 = out["plain"]["hooks"][names[-2]].cpu().numpy()
print("last proj tap max diff", np.abs(a2-b2).max())
print("default", a[0, :6, 13, 31], "\nplain  ", b[0, :6, 13, 31], "\noracle ", taps[names[-1]][0, :6, 13, 31])
print("features default/plain last pixel", out["default"]["features"].cpu().numpy()[0, :4, 13, 31], out["plain"]["features"].cpu().numpy()[0, :4, 13, 31])
