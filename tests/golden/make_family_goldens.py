"""Small-batch known answers for BASELINE configs C3 (EAT 5 s clips) and C5 (EfficientNet-B0 10 s clips) and for AVES (10 s clips), from the NumPy oracles on the
synthetic checkpoints (``avex_amd.synth``):  python tests/golden/make_family_goldens.py  ->  tests/golden/family_small.npz

PARITY UNPINNED: both oracles restate third-party arithmetic that is absent from /root/reference (EAT's remote code, torchvision's
EfficientNet, torchaudio's wav2vec2), so these vectors pin the HIP path to the ORACLE, not to the reference.  ``bench.py`` compares its C3 /
C5 legs with them after the timed region (it may not import ``oracle/`` outside its cpu_baseline leg) and says "unpinned" in the line."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avex_amd import synth  # noqa: E402
from oracle import beats_oracle as BO  # noqa: E402
from oracle import eat_oracle as EO  # noqa: E402
from oracle import effnet_oracle as FO  # noqa: E402
from oracle import aves_oracle as AO  # noqa: E402


def main() -> None:
    out = {}
    t0 = time.time()
    cfg = synth.EAT_BASE_CFG
    wav = synth.noise_clips(2, 80000, seed=12)
    ref, _ = EO.eat_forward(wav, synth.eat_state_dict(cfg), cfg)
    out["eat.seed"] = np.array([12]); out["eat.samples"] = np.array([80000])
    out["eat.pooled_mean"] = ref.mean(1).astype(np.float32)              # [2, 768]: mean over the 513 tokens (eat_hf.py:283-288 "mean")
    out["eat.cls"] = ref[:, 0].astype(np.float32)                        # [2, 768]: the class token, the wrapper's DEFAULT pooling (eat_hf.py:149,281-282)
    print(f"EAT 2 x 5 s: {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    sd = synth.effnet_b0_state_dict()
    x = synth.noise_clips(2, 160000, seed=71)
    mel = BO.audio_processor(x, n_fft=800, hop=160)
    feats, _ = FO.effnet_features(mel, sd, synth.EFFNET_B0_STAGES)
    out["effnet.seed"] = np.array([71]); out["effnet.samples"] = np.array([160000])
    out["effnet.pooled"] = feats.mean((2, 3)).astype(np.float32)         # [2, 1280]: global average pool of the feature map
    print(f"EfficientNet-B0 2 x 10 s: {time.time() - t0:.1f} s, features {feats.shape}", flush=True)
    t0 = time.time()
    acfg = synth.AVES_BASE_CFG
    ax = synth.noise_clips(2, 160000, seed=33)
    aref, _ = AO.aves_forward(ax, synth.aves_state_dict(acfg), acfg)
    out["aves.seed"] = np.array([33]); out["aves.samples"] = np.array([160000])
    out["aves.pooled"] = aref.mean(1).astype(np.float32)                 # [2, 768]: mean over the 499 frames of the last layer (aves_model.py:129-150 features, mean-aggregated)
    out["aves.frame0"] = aref[:, 0].astype(np.float32)                   # [2, 768]: one un-averaged row per clip
    print(f"AVES 2 x 10 s: {time.time() - t0:.1f} s, features {aref.shape}", flush=True)
    np.savez(os.path.join(ROOT, "tests", "golden", "family_small.npz"), **out)


if __name__ == "__main__":
    main()
