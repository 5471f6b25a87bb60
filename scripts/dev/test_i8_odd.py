import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext._types import DenseDev
for n, m in [(50_000, 101), (20_001, 127), (8192, 67)]:
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    X = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    cmax = X.abs().amax(dim=0).contiguous()
    out = xd.dense_sandwich_i8(DenseDev(X, n, m, 0), d, cmax)   # (odd m: supported since round 5)
    want = X.T @ (X * d[:, None])
    dg = torch.sqrt(torch.diagonal(want)); den = torch.outer(dg, dg)
    print(n, m, "err", float(((out - want).abs() / den).max()))
import ctypes as C
from tabmat_amd import _lib
_lib.call("tm_profile_enable", 1)
def t(f, k=4):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
n = 10_000_000
for m in (127, 128, 101):
    g = torch.Generator(device="cuda"); g.manual_seed(m)
    X = torch.randn((n, m), dtype=torch.float64, device="cuda", generator=g)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    cmax = X.abs().amax(dim=0).contiguous()
    blk = DenseDev(X, n, m, 0)
    a = t(lambda: xd.dense_sandwich_i8(blk, d, cmax))
    b = t(lambda: xd.dense_sandwich(blk, d, None, None))
    print(f"10M x {m}: K1e {a:.3f} ms   tm_dense_sandwich_f64 {b:.3f} ms", flush=True)
    del X, blk
