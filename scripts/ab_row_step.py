"""The whole 256-clip BEATs step with the attention output projection on the streaming kernel (default) and on the full-row kernel
(AVEX_AMD_GEMM_ROW=1, read per launch), alternating inside one process:  python scripts/ab_row_step.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avex_amd import kernels as K, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
enc = K.BeatsEncoder(synth.BEATS_BASE_CFG, synth.beats_state_dict(synth.BEATS_BASE_CFG, seed=0), operand_dtype="f16", residual="half")
wav = torch.from_numpy(synth.noise_clips(256, 160000, seed=0)).cuda()


def step_ms(n=10):
    for _ in range(3):
        enc.forward(wav, want_features=False, want_pooled=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = enc.forward(wav, want_features=False, want_pooled=True)["pooled"]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ref = None
for i in range(pairs):
    os.environ.pop("AVEX_AMD_GEMM_ROW", None)
    a, oa = step_ms()
    os.environ["AVEX_AMD_GEMM_ROW"] = "1"
    b, ob = step_ms()
    print(f"pair {i}: streaming {a:7.3f} ms ({256 / a * 1e3:7.0f} clips/s)   full-row out_proj {b:7.3f} ms ({256 / b * 1e3:7.0f} clips/s)   bit-identical embeddings {torch.equal(oa, ob)}", flush=True)
