import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.ext import sparse as xs
n = 10_000_000
dm = synth.dense_block(n, 128, torch.float64, 3)
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
lg = sm._lg()
for _ in range(3):
    xs.csr_dense_sandwich_lg(lg, dm._dev_c(), d, unc=2)
torch.cuda.synchronize()
