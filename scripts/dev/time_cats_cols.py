"""Column-restricted products of a categorical-heavy design (glum's active set), 2M rows, ms."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.split_matrix import SplitMatrix
N = 2_000_000


def tmin(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for k, L, kd in ((20, 50, 0), (24, 10, 32)):
    blocks = [synth.cat_block(N, L, 100 + i) for i in range(k)]
    if kd:
        blocks = [synth.dense_block(N, kd, torch.float64, 3)] + blocks
    X = SplitMatrix(blocks)
    p = X.shape[1]
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    v = torch.rand(p, dtype=torch.float64, device="cuda")
    full = np.arange(p)
    most = np.delete(full, [3, p // 2])
    print(f"{k} cats x {L} + dense {kd}: sandwich all {tmin(lambda: X.sandwich(d)):.3f}  cols=arange {tmin(lambda: X.sandwich(d, cols=full)):.3f}"
          f"  cols=all but 2 {tmin(lambda: X.sandwich(d, cols=most)):.3f} | matvec cols=arange {tmin(lambda: X.matvec(v, cols=full)):.3f}"
          f"  all but 2 {tmin(lambda: X.matvec(v, cols=most)):.3f} | transpose_matvec all but 2 {tmin(lambda: X.transpose_matvec(d, cols=most)):.3f}", flush=True)
