import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
torch.manual_seed(0)
wav = torch.from_numpy(synth.noise_clips(32, 160000, seed=0)).cuda()
plan = K.FbankPlan(norm_mean=15.41663, norm_div=13.11164)
M = 32 * 496
x = torch.randn(M, 768, device="cuda").half(); w = (torch.randn(2304, 768, device="cuda") * 0.05).half()
x2 = torch.randn(M, 768, device="cuda").half(); w2 = (torch.randn(768, 768, device="cuda") * 0.05).half()
x32 = torch.randn(M, 768, device="cuda"); lw = torch.rand(768, device="cuda"); lb = torch.rand(768, device="cuda")
qkv = torch.randn(M, 2304, device="cuda").half(); tab = torch.randn(12, 991, device="cuda")
victims = {"fbank": lambda: plan(wav), "gemm_v2": lambda: K.gemm(x2, w2, out_f32=False, out_half=True, variant=2)["half"],
           "gemm_v3": lambda: K.gemm(x2, w2, out_f32=False, out_half=True, variant=3)["half"],
           "layernorm": lambda: K.layernorm(x32, lw, lb)[0], "attention": lambda: K.attention(qkv, 32, 496, 12, tab, None, None, None),
           "mean_pool": lambda: K.mean_pool(x32.view(32, 496, 768))}
refs = {k: f().clone() for k, f in victims.items()}
torch.cuda.synchronize()
def agg(v): return lambda: K.gemm(x, w, out_f32=False, out_half=True, variant=v)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for an, af in {"gemm_v3": agg(3), "gemm_v1": agg(1), "gemm_v2": agg(2)}.items():
    for vn, vf in victims.items():
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for _ in range(30): af()
        with torch.cuda.stream(sb):
            outs = [vf() for _ in range(5)]
        torch.cuda.synchronize()
        bad = [int((o != refs[vn]).sum()) for o in outs]
        print(f"victim {vn:10s} beside {an:8s}: wrong {bad}")
