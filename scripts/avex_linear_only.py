import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from avex_amd import kernels as K
M = 256 * 496
for N, Kd in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, Kd, device="cuda").half(); w = (torch.randn(N, Kd, device="cuda") * 0.05).half(); b = torch.randn(N, device="cuda")
    for _ in range(5): K.gemm(x, w, bias=b, out_f32=False, out_half=True)
torch.cuda.synchronize()
