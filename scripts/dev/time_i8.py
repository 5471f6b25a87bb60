"""cfg4 dense block (10M x 128 f64): int8-sliced syrk vs the f64 syrk (library event pair around the main kernel)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import dense as xd
n = int(os.environ.get("N", 10_000_000))
dm = synth.dense_block(n, 128, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
Xd = dm._dev_c()
cmax = Xd.as_2d().abs().amax(dim=0).contiguous()
_lib.call("tm_profile_enable", 1)
def t(fn, reps=5):
    ts = []
    for _ in range(reps):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts), out
tf, ref = t(lambda: xd.dense_sandwich(Xd, d, None, None))
print(f"f64 syrk_co : {tf:.3f} ms")
for g in (256, 512):
    _lib.call("tm_tune_set", b"i8_grid", g)
    ti, out = t(lambda: xd.dense_sandwich_i8(Xd, d, cmax))
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"int8 syrk grid {g}: {ti:.3f} ms   max|diff| / max|ref| = {err:.2e}")
