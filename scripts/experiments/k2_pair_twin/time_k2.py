"""K2 at cfg4 size: chunk-major kernel (v3) vs pair-twin kernel (v4).  usage: time_k2.py [rows] [f64|f32]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext._types import PairTwin
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
sm = synth.sparse_block(n, 512, 0.05, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
def timed(fn, reps=5):
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(reps):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    return min(ts), out
A = sm._dev()
t3, ref = timed(lambda: xs.sparse_sandwich_chunked(A, d))
print(f"v3 chunk-major: {t3:.3f} ms", flush=True)
tw = PairTwin.from_csr(A)
print(f"pair twin: base {tw.bv.numel() / A.data.numel():.2f}x nnz slots, overflow {tw.n_ov / A.data.numel():.3f} of nnz", flush=True)
t4, out = timed(lambda: xs.sparse_sandwich_pair(tw, d))
print(f"v4 pair twin:   {t4:.3f} ms   rel.diff {((out - ref).abs().max() / ref.abs().max()).item():.2e}", flush=True)
