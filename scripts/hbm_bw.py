import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
nbytes = 585*1024*1024
a = torch.empty(nbytes//2, dtype=torch.float16, device="cuda"); b = torch.empty_like(a)
ms = t(lambda: a.zero_()); print(f"write-only (zero_) {nbytes/ms/1e9:.2f} TB/s")
ms = t(lambda: a.fill_(1.5)); print(f"write-only (fill_) {nbytes/ms/1e9:.2f} TB/s")
ms = t(lambda: b.copy_(a)); print(f"copy r+w {2*nbytes/ms/1e9:.2f} TB/s total")
ms = t(lambda: a.sum()); print(f"read-only (sum) {nbytes/ms/1e9:.2f} TB/s")
