#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cat > /tmp/stft_t.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from avex_amd import kernels as K
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
wav = (0.1 * torch.randn(256, 160000)).cuda()
for _ in range(3): y = plan(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): y = plan(wav)
torch.cuda.synchronize(); print(f"{os.environ.get('AVEX_AMD_LIB_SUFFIX','product'):8s} melspec 256 x 10 s: {(time.perf_counter()-t0)/30*1e3:.3f} ms")
PY
for sfx in "" _k15 _k31 _k47 _k79 _k127 ""; do AVEX_AMD_LIB_SUFFIX=$sfx AVEX_AMD_LIB=$R/avex_amd/lib/libavexhip$sfx.so python /tmp/stft_t.py 2>&1 | grep -v amdgpu.ids; done
