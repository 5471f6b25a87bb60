"""K2b against the number of waves per workgroup (tm_tune_set "k2b_waves"): does the kernel scale with occupancy?"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
sm = X.matrices[1]
_lib.call("tm_profile_enable", 1)
for waves in (16, 12, 8, 4, 16):
    _lib.call("tm_tune_set", b"k2b_waves", waves)
    ts = []
    for _ in range(4):
        sm._sandwich_dev(d, None, None)
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    print(f"waves {waves:2d}: {min(ts):.3f} ms", flush=True)
