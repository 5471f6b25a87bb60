"""categorical x dense (all categoricals, one pass over the dense block) at cfg4: main-kernel time (tm_profile)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm = X.matrices[0]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in X.matrices[2:]]
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(8):
    out = xsplit.multi_cat_dense_sandwich(cats, d, dm._dev_c())
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"cat x dense: min {min(ts):.3f} ms  all {' '.join('%.3f' % t for t in ts)}   checksum {float(out.sum()):.6e}")
