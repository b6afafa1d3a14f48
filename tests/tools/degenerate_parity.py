import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from avex_amd import synth, kernels as K
from oracle import beats_oracle as O
from _util import rel_l2
sd = synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0)
n = 32000
rng = np.random.default_rng(17)
x = np.zeros((6, n), np.float32)
x[1] = 0.25
x[2, 12345] = 1.0
x[3] = np.where((np.arange(n) // 40) % 2 == 0, 1.0, -1.0)
x[4] = rng.standard_normal(n).astype(np.float32)
x[5, :n // 2] = 0.1 * rng.standard_normal(n // 2).astype(np.float32)
f, _ = O.beats_forward(x, sd, synth.BEATS_BASE_CFG)
want = O.pooled(f)
for dt in ("f16", "bf16"):
    for res in ("f32", "half"):
        for fold in ("0", "1"):
            if fold == "1" and res == "f32": continue
            os.environ["AVEX_AMD_LN_FOLD"] = fold
            enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, sd, operand_dtype=dt, max_chunk_clips=3, residual=res)
            r = enc.forward(torch.from_numpy(x).cuda(), want_features=True, want_pooled=True)
            got = r["pooled"].cpu().numpy(); ft = r["features"].cpu().numpy()
            print(dt, res, "fold" + fold, "pooled:", " ".join(f"{rel_l2(got[i:i+1], want[i:i+1]):.2e}" for i in range(6)),
                  "| frames:", " ".join(f"{rel_l2(ft[i], f[i]):.2e}" for i in range(6)))
            enc.close()
