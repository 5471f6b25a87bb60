"""Tiled K2 (S = 2) vs the direct global-atomic kernel on wide, very sparse blocks (2M rows)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = 2_000_000
def timed(fn):
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(4):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    return min(ts), out
for m, dens in ((2048, 0.0125), (2048, 0.004), (2048, 0.002), (4096, 0.002), (4096, 0.0005), (1024, 0.004)):
    sm = synth.sparse_block(n, m, dens, torch.float64, 7)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    A = sm._dev()
    t1, ref = timed(lambda: xs.sparse_sandwich_chunked(A, d))
    t2, out = timed(lambda: xs.sparse_sandwich_direct(A, d))
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"{m}@{dens * 100:g}% : tiled {t1:8.3f} ms   direct {t2:8.3f} ms   model says direct: {xs.direct_sandwich_pays(A)}   rel.diff {err:.1e}", flush=True)
    del sm, A, ref, out
    torch.cuda.empty_cache()
