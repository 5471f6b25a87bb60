"""K2 at low densities (10M x 512): the per-(group, tile) cost when most lists are empty."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = 10_000_000
for dens in (0.05, 0.005, 0.0005):
    X = synth.mixed_split(n, 8, 512, (8,), dens, torch.float64, 3)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    sm = X.matrices[1]
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(3):
        sm._sandwich_dev(d, None, None)
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    print(f"K2 density {dens}: {min(ts):.3f} ms")
    del X, sm
    torch.cuda.empty_cache()
