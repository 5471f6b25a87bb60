import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
import tabmat_amd as tm
n = 10_000_000
X = synth.mixed_split(n, 64, 512, (8,), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(4):
    sm._cross_sandwich_dev(dm, d, None, None, None)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"K3 narrow (64 dense cols): {min(ts):.3f} ms")
