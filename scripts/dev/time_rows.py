"""sandwich(d, rows=...) at 2M rows: masked-d fast paths vs the row-list kernels, per block op."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
import tabmat_amd as tm
from tabmat_amd.ext import sparse as xs, dense as xd, split as xsplit
n = 2_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
X.to_device()
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
cats = X.matrices[2:]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print(f"full sandwich: {t(lambda: X.sandwich(d)):.2f} ms")
for frac in (0.5, 0.1, 0.01):
    rows = torch.sort(torch.randperm(n, device="cuda")[: int(n * frac)]).values.to(torch.int32)
    print(f"--- rows = {frac:.0%} of n")
    print(f"  SplitMatrix.sandwich(rows)         {t(lambda: X._sandwich_dev(d, rows, None)):8.2f} ms")
    print(f"  dense self   (row-list syrk)       {t(lambda: dm._sandwich_dev(d, rows, None)):8.2f} ms")
    print(f"  sparse self  masked chunked        {t(lambda: sm._sandwich_dev(d, rows, None)):8.2f} ms")
    print(f"  sparse self  row-list generic      {t(lambda: xs.sparse_sandwich(sm._dev(), d, rows, None)):8.2f} ms")
    print(f"  sparse x dense masked fast         {t(lambda: sm._cross_sandwich_dev(dm, d, rows, None, None)):8.2f} ms")
    print(f"  sparse x dense row-list generic    {t(lambda: xs.csr_dense_sandwich(sm._dev(), dm._dev_c(), d, rows, None, None)):8.2f} ms")
    c0 = cats[0]
    print(f"  cat x dense  masked fused          {t(lambda: c0._cross_sandwich_dev(dm, d, rows, None, None)):8.2f} ms")
    print(f"  cat x dense  row-list              {t(lambda: xsplit.sandwich_cat_dense(c0._dev(), c0.shape[1], d, dm._dev(), rows, None, False)):8.2f} ms")
    print(f"  cat x sparse row-list              {t(lambda: xsplit.sandwich_cat_sparse(c0._dev(), c0.shape[1], d, sm._dev(), rows, None, False)):8.2f} ms")
    print(f"  device row indexing X[rows]        {t(lambda: X[rows.to(torch.int64).cpu().numpy()], 1):8.2f} ms")
