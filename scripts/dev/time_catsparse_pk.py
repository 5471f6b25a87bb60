"""categorical x sparse on the entry twin at cfg4: codes gathered per categorical vs packed into one word per row."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
sm = X.matrices[1]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in X.matrices[2:]]
ent = sm._ent()
pk = xsplit.pack_codes(cats)
_lib.call("tm_profile_enable", 1)
def t(f, k=5):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
for r in range(3):
    a = t(lambda: xsplit.multi_cat_sparse_sandwich_ent(cats, d, ent))
    b = t(lambda: xsplit.multi_cat_sparse_sandwich_ent(cats, d, ent, pk))
    print(f"codes per categorical {a:.3f} ms   packed {b:.3f} ms", flush=True)
