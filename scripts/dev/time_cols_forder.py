"""Narrow selection with an F-ordered dense block (pandas-style input), cfg4 recipe, ms."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd as tm
from tabmat_amd import synth
N = int(os.environ.get("TM_ROWS", "2000000"))
X = synth.mixed_split(N)
blocks = list(X.matrices)
t = blocks[0]._dev().as_2d()
blocks[0] = tm.DenseMatrix(t.T.contiguous().T)          # same values, column-major in HBM
X = tm.SplitMatrix(blocks, X.indices)
p = X.shape[1]
d = torch.rand(N, dtype=torch.float64, device="cuda")
rng = np.random.default_rng(0)
for share in (1.0, 0.1, 0.05, 0.01):
    cols = None if share == 1.0 else np.sort(rng.choice(p, int(share * p), replace=False))
    for _ in range(2):
        X.sandwich(d, cols=cols)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); X.sandwich(d, cols=cols); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"F-ordered dense block, {share:4.2f} of the columns: sandwich {min(ts) * 1e3:.3f} ms", flush=True)
