import sys, os
sys.path.insert(0, os.getcwd())
import torch
from avex_amd import kernels as K, synth
x = torch.from_numpy(synth.noise_clips(256, 160000, seed=0)).cuda()
plan = K.FbankPlan()
for _ in range(3): plan(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan(x)
e1.record(); torch.cuda.synchronize()
print(f"fbank f32 out 256 clips: {e0.elapsed_time(e1)/20*1e3:.1f} us")
