"""rows = 10 % of 2M: the kernels of one row-restricted sandwich (run under rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
n = 2_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
X.to_device()
d = torch.rand(n, dtype=torch.float64, device="cuda")
rows = torch.sort(torch.randperm(n, device="cuda")[: n // 10]).values.to(torch.int32)
for _ in range(6):
    X._sandwich_dev(d, rows, None)
torch.cuda.synchronize()
