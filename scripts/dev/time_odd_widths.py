"""Odd-width dense blocks (4M rows, f64): the dispatched tm_dense_sandwich_f64 (narrow / generic element-load syrk below
65 columns) against K1c forced, and K1e where it applies."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import _lib
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext._types import DenseDev
_lib.call("tm_profile_enable", 1)
def t(f, k=4):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
n = 4_000_000
for m in ([int(a) for a in sys.argv[1:]] or [13, 17, 31, 33, 47, 63, 64, 65, 99, 127]):
    g = torch.Generator(device="cuda"); g.manual_seed(m)
    X = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    blk = DenseDev(X, n, m, 0)
    a = t(lambda: xd.dense_sandwich(blk, d, None, None))
    b = t(lambda: xd.dense_sandwich_co(blk, d))
    print(f"4M x {m:3d}: dispatched {a:.3f} ms   K1c {b:.3f} ms", flush=True)
    del X, blk
