"""cfg4 recipe at 2M rows: sandwich / matvec / transpose_matvec with column selections (ms)."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
N = int(os.environ.get("TM_ROWS", "2000000"))


def tmin(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


X = synth.mixed_split(N)
p = X.shape[1]
d = torch.rand(N, dtype=torch.float64, device="cuda")
v = torch.rand(p, dtype=torch.float64, device="cuda")
rng = np.random.default_rng(0)
print(f"all columns: sandwich {tmin(lambda: X.sandwich(d)):.3f}  matvec {tmin(lambda: X.matvec(v)):.3f}  transpose_matvec {tmin(lambda: X.transpose_matvec(d)):.3f}")
for share in (0.99, 0.75, 0.5, 0.4, 0.35, 0.3, 0.25, 0.22, 0.2, 0.18, 0.15, 0.1, 0.05, 0.01):
    cols = np.sort(rng.choice(p, int(share * p), replace=False))
    nc = int(np.sum(cols < 640))
    print(f"{share:4.2f} of the columns ({nc:3d} dense + sparse): sandwich {tmin(lambda: X.sandwich(d, cols=cols)):.3f}  matvec {tmin(lambda: X.matvec(v, cols=cols)):.3f}"
          f"  transpose_matvec {tmin(lambda: X.transpose_matvec(d, cols=cols)):.3f}", flush=True)
