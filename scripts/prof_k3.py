import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tabmat_amd import synth
n = 2_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
for _ in range(3):
    sm._cross_sandwich_dev(dm, d, None, None, None)
torch.cuda.synchronize()
