"""K2e: packed 12-byte records against the 16-byte ones, interleaved, with a parity check (experiment)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 2_000_000))
_lib.call("tm_profile_enable", 1)
def t(f, k=5):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
for m, dens, dt in ((2048, 0.0125, torch.float64), (4096, 0.00625, torch.float64), (4096, 0.002, torch.float64), (2048, 0.0125, torch.float32)):
    sm = synth.sparse_block(n, m, dens, dt, 1003)
    d = torch.rand(n, dtype=dt, device="cuda"); d[::9] = 0
    A = sm._dev()
    res = {}
    for rnd in range(2):
        for pk in (False, True):
            xs.K2_PAIRS_PACKED = pk
            out = xs.sparse_sandwich_pairs(A, d)
            res.setdefault(pk, []).append(t(lambda: xs.sparse_sandwich_pairs(A, d)))
            res[("o", pk)] = out
    err = float((res[("o", False)] - res[("o", True)]).abs().max() / res[("o", False)].abs().max())
    print(f"m={m} dens={dens} {str(dt)[6:]}: 16-byte {res[False]}  packed {res[True]}  rel diff {err:.1e}", flush=True)
    del sm, A, res
    torch.cuda.empty_cache()
