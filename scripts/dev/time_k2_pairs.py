"""Sparse self sandwich of wide blocks: the pair-stream kernel (tm_sparse_sandwich_pairs_*) against what the
dispatch picks today, with a correctness check against the generic kernel."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 2_000_000))
cases = [(512, 0.05), (1024, 0.025), (2048, 0.0125), (4096, 0.002), (4096, 0.00625), (2048, 0.05), (8192, 0.0005)]
if len(sys.argv) > 1:
    cases = [(int(sys.argv[1]), float(sys.argv[2]))]
_lib.call("tm_profile_enable", 1)
def t(f, k=4):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
for m, dens in cases:
    sm = synth.sparse_block(n, m, dens, torch.float64, 1003)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    A = sm._dev()
    auto = xs.pairs_sandwich_pays(A)
    xs.K2_PAIRS = "0"                  # what the dispatch takes without the pair-stream kernel
    ref = sm._sandwich_dev(d, None, None)
    t_cur = t(lambda: sm._sandwich_dev(d, None, None))
    xs.K2_PAIRS = "auto"
    got = xs.sparse_sandwich_pairs(A, d)
    dg = torch.sqrt(torch.diagonal(ref).abs()); den = torch.outer(dg, dg).clamp_min(1e-300)
    err = float(((got - ref).abs() / den).max())
    t_new = t(lambda: xs.sparse_sandwich_pairs(A, d))
    nnz = A.data.numel()
    print(f"m={m:5d} dens={dens:.5f} nnz/row={nnz / n:5.1f}  without {t_cur:7.3f} ms   pairs {t_new:7.3f} ms   model picks pairs: {auto}   nat.err {err:.1e}", flush=True)
    del sm, A, ref, got
    torch.cuda.empty_cache()
