import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, ctypes
from avex_amd import kernels as K, synth
x = torch.from_numpy(synth.noise_clips(8, 160000, seed=0)).cuda()
y = K.FbankPlan()(x).cpu().numpy()
np.save(sys.argv[1], y)
