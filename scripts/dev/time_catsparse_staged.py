"""categorical x sparse on the entry twin at cfg4: per-slot gathers of {d, code word} (round 5) vs the rows' operands
staged per slab in per-wave LDS (round 6, tm_tune_set catsparse_staged); results compared entry by entry."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit
n = int(os.environ.get("N", 10_000_000))
M, DENS = int(os.environ.get("M", 512)), float(os.environ.get("DENS", 0.05))
X = synth.mixed_split(n, 128, M, (256, 96, 32), DENS, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
d[::13] = 0
sm = X.matrices[1]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in X.matrices[2:]]
ent = sm._ent()
pk = xsplit.pack_codes(cats)
_lib.call("tm_profile_enable", 1)
def t(f, k=5):
    ts = []
    for _ in range(k):
        r = f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts), r
for rep in range(3):
    out = {}
    for staged in (0, 1):
        _lib.call("tm_tune_set", b"catsparse_staged", staged)
        _lib.call("tm_tune_set", b"catsparse_staged_fill", 0)
        a, ra = t(lambda: xsplit.multi_cat_sparse_sandwich_ent(cats, d, ent))
        b, rb = t(lambda: xsplit.multi_cat_sparse_sandwich_ent(cats, d, ent, pk))
        out[staged] = (a, b, ra, rb)
    e1 = float((out[0][2] - out[1][2]).abs().max() / out[0][2].abs().max())
    e2 = float((out[0][3] - out[1][3]).abs().max() / out[0][3].abs().max())
    print(f"{M} cols @ {DENS}: {ent.n_slots() / (ent.bstart.shape[0] * (ent.bstart.shape[1] - 1)):.1f} slots / block  "
          f"gathers: codes {out[0][0]:.3f} packed {out[0][1]:.3f} ms   staged: codes {out[1][0]:.3f} packed {out[1][1]:.3f} ms"
          f"   max rel diff {e1:.1e} {e2:.1e}", flush=True)
