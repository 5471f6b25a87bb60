"""A/B in one process: categorical diagonals from the pair tables vs their own histogram passes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
import tabmat_amd.split_matrix as sm
n = 10_000_000
X = synth.mixed_split(n)
d = torch.rand(n, dtype=torch.float64, device="cuda")
def best(k=12):
    for _ in range(3):
        X.sandwich(d)
    torch.cuda.synchronize()
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); X.sandwich(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[k // 2] * 1e3
for rep in range(2):
    for flag in (True, False):
        sm.DIAG_FROM_PAIRS = flag
        print(f"diag from pairs {flag!s:5s}: min {best()[0]:.3f}  median {best()[1]:.3f} ms", flush=True)
