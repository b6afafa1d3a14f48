import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from avex_amd import kernels as K, _capi
M = 32 * 496
x = torch.randn(M, 768, device="cuda").half(); w = (torch.randn(2304, 768, device="cuda") * 0.05).half()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for v in (1, 3, 2, 0):
    rep = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(30):
            if v: K.gemm(x, w, out_f32=False, out_half=True, variant=v)
    with torch.cuda.stream(sb):
        for _ in range(5):
            _capi.lib().avexhip_debug_lds_canary(4000, 200, rep.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = rep.cpu().tolist()
    print(f"canary beside gemm variant {v}: mismatches={r[0]} first_word={r[1]} value={r[2] & 0xffffffff:#x} block={r[3]}")
