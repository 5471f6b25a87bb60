"""sandwich(d, rows=...) at 2M rows: the row-list kernels against the full pass."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
import tabmat_amd as tm
from tabmat_amd.ext import sparse as xs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
X.to_device()
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
full = t(lambda: X.sandwich(d))
print(f"full sandwich: {full:.2f} ms")
for frac in (0.5, 0.25, 0.1, 0.01):
    rows = torch.sort(torch.randperm(n, device="cuda")[: int(n * frac)]).values.to(torch.int32)
    tt = t(lambda: X._sandwich_dev(d, rows, None))
    print(f"--- rows = {frac:.0%} of n: SplitMatrix.sandwich(rows) {tt:7.2f} ms  ({full / tt:.2f}x faster than full)")
    dmask = torch.zeros_like(d); dmask[rows.long()] = d[rows.long()]
    print(f"  sparse self  rows {t(lambda: xs.sparse_sandwich_rows(sm._dev(), d, rows)):7.2f}   masked {t(lambda: xs.sparse_sandwich_chunked(sm._dev(), dmask)):7.2f}")
    print(f"  sparse x dense rows {t(lambda: xs.csr_dense_sandwich_rows(sm._dev(), dm._dev_c(), d, rows)):7.2f}   masked {t(lambda: xs.csr_dense_sandwich_lg(sm._lg(), dm._dev_c(), dmask)):7.2f}")
    print(f"  dense self    {t(lambda: dm._sandwich_dev(d, rows, None)):7.2f}")
