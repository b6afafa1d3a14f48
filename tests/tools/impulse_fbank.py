import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from avex_amd import kernels as K
from oracle import beats_oracle as O
n = 32000
x = np.zeros((1, n), np.float32); x[0, 12345] = 1.0
want = O.fbank(x * np.float32(2 ** 15))[0]
got = K.FbankPlan()(torch.from_numpy(x).cuda()).cpu().numpy()[0]
d = np.abs(got - want)
fr = np.where(d.max(1) > 1e-4)[0]
print("frames with |diff| > 1e-4:", fr, "impulse frames:", (12345 - 400) // 160 + 1, "..", 12345 // 160)
for t in fr[:6]:
    print(t, "max diff", d[t].max(), "at mel", d[t].argmax(), "want", want[t, ::16].round(3), "got", got[t, ::16].round(3))
# the same frames in float64
w = O.povey_window(400).astype(np.float64)
for t in fr[:3]:
    seg = (x[0, t * 160:t * 160 + 400] * 32768.0).astype(np.float64)
    seg = seg - seg.mean()
    seg = seg - 0.97 * np.concatenate([seg[:1], seg[:-1]])
    sp = np.abs(np.fft.rfft(np.pad(seg * w, (0, 112)))) ** 2
    mel = sp @ O.mel_filterbank().astype(np.float64)
    ref64 = np.log(np.maximum(mel, 1.1920929e-07))
    print(t, "vs float64: oracle err", np.abs(want[t] - ref64).max(), "gpu err", np.abs(got[t] - ref64).max())
