"""Chunked K2 kernel: byte columns against int32 columns, interleaved (2M rows)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 2_000_000))
_lib.call("tm_profile_enable", 1)
def t(f, k=6):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
for m, dens in ((512, 0.05), (512, 0.01), (512, 0.002), (1024, 0.025), (256, 0.05)):
    sm = synth.sparse_block(n, m, dens, torch.float64, 1003)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    A = sm._dev()
    res = {}
    for rnd in range(2):
        for u8 in (False, True):
            xs.K2B_U8 = u8
            out = xs.sparse_sandwich_chunked(A, d)
            res.setdefault(u8, []).append(round(t(lambda: xs.sparse_sandwich_chunked(A, d)), 3))
            res[("o", u8)] = out
    diff = float((res[("o", False)] - res[("o", True)]).abs().max() / res[("o", False)].abs().max())
    print(f"m={m} dens={dens}: int32 {res[False]}  bytes {res[True]}  rel diff {diff:.1e}", flush=True)
    del sm, A, res
    torch.cuda.empty_cache()
