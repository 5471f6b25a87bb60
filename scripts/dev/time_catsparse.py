"""Fused cat x sparse (multi_cat_sparse_pf_kernel) at cfg4 size."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit
n = 10_000_000
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
cats = [synth.cat_block(n, c, 2000 + i) for i, c in enumerate((256, 96, 32))]
d = torch.rand(n, dtype=torch.float64, device="cuda")
cl = [(c._dev(), c.shape[1], c.drop_first) for c in cats]
slab = sm._slab()
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(6):
    out = xsplit.multi_cat_sparse_sandwich(cl, d, slab)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"cat x sparse: min {min(ts):.3f} ms  median {sorted(ts)[3]:.3f}", flush=True)
