"""Categorical-only designs (many glum models are mostly categorical): k categoricals of L levels,
2M rows: SplitMatrix.sandwich ms, and what the pair tables cost."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.split_matrix import SplitMatrix
N = 2_000_000
import sys as _s
CASES = ((10, 100), (20, 50), (6, 2000), (30, 12), (16, 30), (8, 40))
if len(_s.argv) > 1:
    CASES = (CASES[int(_s.argv[1])],)
for k, L in CASES:
    X = SplitMatrix([synth.cat_block(N, L, 100 + i) for i in range(k)])
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    for _ in range(2):
        X.sandwich(d)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); X.sandwich(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    pairs = k * (k - 1) // 2
    print(f"{k:3d} categoricals x {L:5d} levels: {min(ts) * 1e3:8.3f} ms  ({pairs} pair tables, {min(ts) * 1e6 / pairs:6.1f} us each; "
          f"codes + d once = {N * (4 * k + 8) / 1e9:.2f} GB)", flush=True)
