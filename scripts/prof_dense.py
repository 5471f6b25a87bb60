import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth
n = 4_000_000
for dt, k in ((torch.float64, 128), (torch.float32, 256)):
    X = synth.dense_block(n, k, dt, 1)
    d = torch.rand(n, dtype=dt, device="cuda")
    for _ in range(3):
        X._sandwich_dev(d, None, None)
torch.cuda.synchronize()
