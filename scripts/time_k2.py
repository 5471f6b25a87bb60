"""Time the unrestricted sparse self sandwich (K2) at cfg4 size."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth, _lib
n = 10_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
sm = X.matrices[1]
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(4):
    sm._sandwich_dev(d, None, None)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"K2: {min(ts):.3f} ms")
