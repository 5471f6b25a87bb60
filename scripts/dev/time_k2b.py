"""K2b (sparse self sandwich on the static block list) at cfg4: main-kernel time (tm_profile), default configuration."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
g = torch.Generator(device="cuda").manual_seed(1)
d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
d[::17] = 0
A = sm._dev()
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(10):
    out = xs.sparse_sandwich_blocks(A, d)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"K2b: min {min(ts):.3f} ms  all {' '.join('%.3f' % t for t in ts)}  checksum {float(out.sum()):.10e}")
