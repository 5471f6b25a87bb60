"""cat x sparse cross terms at cfg4 size: slab-form kernel against the entry-twin kernel (round 4).
usage: python scripts/dev/time_catsparse_ent.py [rows]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt = torch.float64
sm = synth.sparse_block(n, 512, 0.05, dt, 1003)
cms = [synth.cat_block(n, c, 2000 + i) for i, c in enumerate((256, 96, 32))]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in cms]
d = torch.rand(n, dtype=dt, device="cuda")
def timed(fn, reps=4):
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(reps):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    return min(ts), out
slab = sm._slab()
t0, ref = timed(lambda: xsplit.multi_cat_sparse_sandwich(cats, d, slab))
ref = ref.clone()
print(f"slab form : {t0:.3f} ms", flush=True)
ent = sm._ent()
for w in (16, 12, 8):
    _lib.call("tm_tune_set", b"catsparse_waves", w)
    t1, out = timed(lambda: xsplit.multi_cat_sparse_sandwich_ent(cats, d, ent))
    print(f"entry twin, {w} waves: {t1:.3f} ms  rel.diff {((out - ref).abs().max() / ref.abs().max()).item():.1e}", flush=True)
