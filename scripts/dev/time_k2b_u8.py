"""K2b at cfg4: byte columns (tm_sparse_sandwich_blocks_u8_*) against int32 columns, interleaved on one box."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
A = sm._dev()
_lib.call("tm_profile_enable", 1)
res = {}
for rnd in range(3):
    for u8 in (False, True):
        xs.K2B_U8 = u8
        ts = []
        for _ in range(5):
            out = xs.sparse_sandwich_blocks(A, d)
            ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
        res.setdefault(u8, []).append(min(ts))
        if rnd == 0:
            res[("out", u8)] = out
print("int32 columns", ["%.3f" % t for t in res[False]], "  byte columns", ["%.3f" % t for t in res[True]],
      "  max abs diff", float((res[("out", False)] - res[("out", True)]).abs().max()))
