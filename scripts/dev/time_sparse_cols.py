"""SparseMatrix.sandwich with narrow column selections (standalone sparse designs), 2M rows, ms."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
import tabmat_amd.sparse_matrix as spm
N = 2_000_000


def tmin(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for m, dens in ((512, 0.05), (2048, 0.0125), (8192, 0.0005)):
    X = synth.sparse_block(N, m, dens, torch.float64, 7)
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    rng = np.random.default_rng(0)
    line = f"{m} columns @ {dens * 100:g} %: all {tmin(lambda: X.sandwich(d)):.3f}"
    for w in (10, 50, 128):
        cols = np.sort(rng.choice(m, w, replace=False))
        a = tmin(lambda: X.sandwich(d, cols=cols))
        old, spm.NARROW_COLS = spm.NARROW_COLS, 0
        b = tmin(lambda: X.sandwich(d, cols=cols))
        spm.NARROW_COLS = old
        line += f" | {w} cols: {b:.3f} -> {a:.3f}"
    print(line, flush=True)
