"""K3 at cfg4 size: round-1 wide-ELL kernel vs the lane-group kernel (unc = 2, 4); checks that
both agree.  usage: python scripts/dev/time_k3_lg.py [rows]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
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
ell = sm._ell(wide=True)
t0, ref = timed(lambda: xs.csr_dense_sandwich_ell(ell, Bd, d))
print(f"ellw: {t0:.3f} ms  slots {ell.vals.numel()/sm._dev().data.numel():.2f}x nnz", flush=True)
del ell; sm._ellwblk = None
lg = sm._lg()
print(f"lg twin: round0 slots {lg.vals.numel()/sm._dev().data.numel():.2f}x nnz, extra rounds {lg.xptr[-1].item()} "
      f"of {lg.xptr.numel()-1} blocks, unc {lg.unc}", flush=True)
for unc in (2, 4):
    t, out = timed(lambda: xs.csr_dense_sandwich_lg(lg, Bd, d, unc=unc))
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"lg unc={unc}: {t:.3f} ms  rel.diff vs ellw {err:.2e}", flush=True)
