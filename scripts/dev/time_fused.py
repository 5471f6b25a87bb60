"""Fused K3 + syrk (one pass over the dense block) vs the two separate kernels at cfg4 size.
usage: python scripts/dev/time_fused.py [rows]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs, dense as xd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt = torch.float64
dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, 0.05, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
def timed(fn, reps=4):
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(reps):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    return min(ts), out
lg = sm._lg()
t_lg, ref_x = timed(lambda: xs.csr_dense_sandwich_lg(lg, Bd, d))
t_sy, ref_s = timed(lambda: xd.dense_sandwich(Bd, d, None, None))
print(f"separate: lg {t_lg:.3f} ms + syrk {t_sy:.3f} ms = {t_lg + t_sy:.3f} ms", flush=True)
t_f, (out_x, out_s) = timed(lambda: xs.csr_dense_sandwich_lg_syrk(lg, Bd, d))
ex = ((out_x - ref_x).abs().max() / ref_x.abs().max()).item()
es = ((out_s - ref_s).abs().max() / ref_s.abs().max()).item()
print(f"fused: {t_f:.3f} ms   rel.diff cross {ex:.2e}  self {es:.2e}", flush=True)
