"""matvec / transpose_matvec of categorical-heavy designs, 2M rows (ms)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.split_matrix import SplitMatrix
N = 2_000_000


def tmin(f, reps=7):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


for k, L, kd in ((20, 50, 0), (30, 12, 0), (24, 10, 32), (3, 100, 128)):
    blocks = [synth.cat_block(N, L, 100 + i) for i in range(k)]
    if kd:
        blocks = [synth.dense_block(N, kd, torch.float64, 3)] + blocks
    X = SplitMatrix(blocks)
    p = X.shape[1]
    v = torch.rand(p, dtype=torch.float64, device="cuda")
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    gb = (N * 4 * k + N * 8 * (kd + 1)) / 1e9
    a = tmin(lambda: X.matvec(v))
    b = tmin(lambda: X.transpose_matvec(d))
    print(f"{k:3d} cats x {L:4d} levels + dense {kd:3d}: matvec {a:7.3f} ms ({gb / a * 1e3:6.0f} GB/s)   "
          f"transpose_matvec {b:7.3f} ms ({gb / b * 1e3:6.0f} GB/s)", flush=True)
