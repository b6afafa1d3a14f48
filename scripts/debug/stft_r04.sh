#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x -k "melspec or stft" 2>&1 | tail -4
cat > /tmp/stft_t.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from avex_amd import kernels as K
plan = K.MelspecPlan(n_fft=800, hop_length=160, n_mels=128, normalize=True)
wav = (0.1 * torch.randn(256, 160000)).cuda()
for _ in range(3): y = plan(wav)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): y = plan(wav)
torch.cuda.synchronize(); print(f"melspec 256 x 10 s: {(time.perf_counter()-t0)/20*1e3:.3f} ms  checksum {float(y.double().sum()):.6f}")
PY
AVEX_AMD_STFT_GENERIC=1 python /tmp/stft_t.py 2>&1 | grep -v amdgpu.ids
python /tmp/stft_t.py 2>&1 | grep -v amdgpu.ids
AVEX_AMD_STFT_GENERIC=1 python /tmp/stft_t.py 2>&1 | grep -v amdgpu.ids
python /tmp/stft_t.py 2>&1 | grep -v amdgpu.ids
