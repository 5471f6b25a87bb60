"""K2 alone at cfg4 size (10M x 512 @ 5 %, f64): kernel time from the library's event pair."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = 10_000_000
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
A = sm._dev()
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(5):
    out = xs.sparse_sandwich_chunked(A, d)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"K2 min {min(ts):.3f} ms  median {sorted(ts)[2]:.3f}  checksum {out.sum().item():.8e}")
