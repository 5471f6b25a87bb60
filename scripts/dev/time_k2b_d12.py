"""K2b at cfg4: 12-byte block descriptors (tm_sparse_sandwich_blocks_p12_*, round 6) against the 16-byte list, both on
byte columns, interleaved on one box; bytes of the two lists."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs, _types as T
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
A = sm._dev()
_lib.call("tm_profile_enable", 1)
res = {}
for rnd in range(3):
    for d12 in (False, True):
        T.K2B_DESC12 = d12
        ts = []
        for _ in range(5):
            out = xs.sparse_sandwich_blocks(A, d)
            ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
        res.setdefault(d12, []).append(min(ts))
        if rnd == 0:
            res[("out", d12)] = out
b16, b12 = A._pb[False][0], A._pb[True][0]
print("16-byte descriptors", ["%.3f" % t for t in res[False]], "  12-byte", ["%.3f" % t for t in res[True]],
      "  max abs diff / max", float((res[("out", False)] - res[("out", True)]).abs().max() / res[("out", True)].abs().max()),
      f"  list bytes {b16.numel() * 4 / 1e9:.2f} -> {b12.numel() * 4 / 1e9:.2f} GB")
