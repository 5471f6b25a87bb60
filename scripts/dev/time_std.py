"""cfg4 design: SplitMatrix.sandwich vs StandardizedMatrix.sandwich (device d), with / without a complete categorical."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd as tm
from tabmat_amd import synth
n = int(os.environ.get("N", 10_000_000))
def t(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
for drop in (False, True):
    blocks = [synth.dense_block(n, 128, torch.float64, 3), synth.sparse_block(n, 512, 0.05, torch.float64, 1003)]
    blocks += [synth.cat_block(n, c, seed=7 + i, drop_first=drop) for i, c in enumerate((256, 96, 32))]
    mat = tm.SplitMatrix(blocks)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    w = np.full(n, 1.0 / n)
    std, _, _ = mat.standardize(w, True, True)
    print(f"drop_first={drop}: plain {t(lambda: mat.sandwich(d)):.2f} ms   standardized {t(lambda: std.sandwich(d)):.2f} ms")
    del mat, std, blocks
    torch.cuda.empty_cache()
