"""Reference design dense_cat (3M x (1000 + 1000 levels + 5 dense)): the cat x dense term through the generic
LDS-tile kernel of tm_multi_cat_dense_sandwich_* against the path SplitMatrix takes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.ext import split as xsplit
m = synth.reference_design(os.environ.get("DESIGN", "dense_cat"))
d = torch.rand(m.shape[0], dtype=torch.float64, device="cuda")
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r
full, _ = t(lambda: m.sandwich(d))
cats = [(c._dev(), c.shape[1], c.drop_first) for c in m.matrices[:2]]
dm = m.matrices[2]
a, ra = t(lambda: xsplit.multi_cat_dense_sandwich(cats, d, dm._dev_c()))
b, rb = t(lambda: m._fused_cats(dm, cats, [0, 1], d, None, 2000, 10**9))
print(f"sandwich {full:.3f} ms | cat x dense: generic tile kernel {a:.3f} ms, current path {b:.3f} ms, "
      f"max diff {(ra - rb).abs().max().item():.2e}")
