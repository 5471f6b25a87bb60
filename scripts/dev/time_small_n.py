"""cfg4 recipe at small row counts: where the launch overhead floor of sandwich / matvec sits (ms)."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth


def tmin(f, reps=9):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for n in (1_000, 10_000, 100_000, 1_000_000):
    X = synth.mixed_split(n)
    p = X.shape[1]
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    v = torch.rand(p, dtype=torch.float64, device="cuda")
    g = X.sandwich_graph(d)
    print(f"n = {n:8d}: sandwich {tmin(lambda: X.sandwich(d)):.3f}  (graph replay {tmin(lambda: g(d)):.3f})  matvec {tmin(lambda: X.matvec(v)):.3f}"
          f"  transpose_matvec {tmin(lambda: X.transpose_matvec(d)):.3f}", flush=True)
