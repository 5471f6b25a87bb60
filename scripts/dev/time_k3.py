"""K3 at cfg4 (10M x 512 sparse @ 5 % against 128 dense f64 columns): main-kernel time from the library's event pair."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 10_000_000))
dm = synth.dense_block(n, 128, torch.float64, 3)
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
lg = sm._lg()
_lib.call("tm_profile_enable", 1)
_lib.call("tm_tune_set", b"lg_rounds", int(os.environ.get("ROUNDS", 1)))
ts = []
for _ in range(6):
    xs.csr_dense_sandwich_lg(lg, dm._dev_c(), d, unc=int(os.environ.get("UNC", 2)))
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"K3 lg f64: min {min(ts):.3f} ms  (all: {' '.join(f'{t:.2f}' for t in ts)})")
