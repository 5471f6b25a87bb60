"""Run each block op of the cfg4 sandwich a few times (for rocprofv3 PMC / kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth
from tabmat_amd.ext import split as xsplit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in X.matrices[2:]]
for _ in range(2):
    dm._sandwich_dev(d, None, None)
    sm._sandwich_dev(d, None, None)
    sm._cross_sandwich_dev(dm, d, None, None, None)
    xsplit.multi_cat_dense_sandwich(cats, d, dm._dev())
    xsplit.multi_cat_sparse_sandwich(cats, d, sm._slab())
torch.cuda.synchronize()
