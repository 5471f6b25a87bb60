"""cfg4 design: matvec / transpose_matvec / sandwich, plain SplitMatrix against StandardizedMatrix (device operands)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd as tm
from tabmat_amd import synth
n = int(os.environ.get("N", 10_000_000))
def t(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
mat = synth.mixed_split(n)
d = torch.rand(n, dtype=torch.float64, device="cuda")
v = torch.rand(mat.shape[1], dtype=torch.float64, device="cuda")
w = np.full(n, 1.0 / n)
std, _, _ = mat.standardize(w, True, True)
for name, f, g in (("sandwich", lambda: mat.sandwich(d), lambda: std.sandwich(d)),
                   ("matvec", lambda: mat.matvec(v), lambda: std.matvec(v)),
                   ("transpose_matvec", lambda: mat.transpose_matvec(d), lambda: std.transpose_matvec(d))):
    print(f"{name:18s} plain {t(f):7.3f} ms   standardized {t(g):7.3f} ms", flush=True)
