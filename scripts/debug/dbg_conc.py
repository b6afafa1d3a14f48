import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from avex_amd import synth, kernels as K
torch.manual_seed(0)
wav = torch.from_numpy(synth.noise_clips(32, 160000, seed=0)).cuda()
plan = K.FbankPlan(norm_mean=15.41663, norm_div=13.11164)
ref = plan(wav).clone(); torch.cuda.synchronize()
M = 32 * 496
x = torch.randn(M, 768, device="cuda").half(); w = (torch.randn(2304, 768, device="cuda") * 0.05).half()
x32 = torch.randn(M, 768, device="cuda"); lw = torch.ones(768, device="cuda"); lb = torch.zeros(768, device="cuda")
qkv = torch.randn(M, 2304, device="cuda").half()
tab = torch.randn(12, 991, device="cuda")
def k_gemm(v): return lambda: K.gemm(x, w, out_f32=False, out_half=True, variant=v)
def k_ln(): K.layernorm(x32, lw, lb)
def k_lnh(): K.layernorm(x, lw, lb)
def k_att(): K.attention(qkv, 32, 496, 12, tab, None, None, None)
def k_pool(): K.mean_pool(x32.view(32, 496, 768))
kernels = {"gemm_v2": k_gemm(2), "gemm_v3": k_gemm(3), "gemm_v1": k_gemm(1), "layernorm_f32": k_ln, "layernorm_half": k_lnh, "attention": k_att, "mean_pool": k_pool, "nothing": lambda: None}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in kernels.items():
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(30): fn()
    with torch.cuda.stream(sb):
        outs = [plan(wav) for _ in range(5)]
    torch.cuda.synchronize()
    bad = [int((o != ref).sum()) for o in outs]
    print(f"fbank beside {name:15s}: wrong elements per run {bad}")
