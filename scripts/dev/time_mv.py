"""matvec / transpose_matvec of the cfg4 blocks and the cfg3 histogram, ms (min of 8)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
def best(fn, k=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
n = 10_000_000
X = synth.mixed_split(n)
v = torch.rand(X.shape[1], dtype=torch.float64, device="cuda")
w = torch.rand(n, dtype=torch.float64, device="cuda")
print(f"split matvec {best(lambda: X.matvec(v)):.3f}  transpose_matvec {best(lambda: X.transpose_matvec(w)):.3f}")
sp = X.matrices[1]
vs = torch.rand(512, dtype=torch.float64, device="cuda")
print(f"sparse matvec {best(lambda: sp.matvec(vs)):.3f}  transpose_matvec {best(lambda: sp.transpose_matvec(w)):.3f}")
del X, sp
torch.cuda.empty_cache()
c = synth.cat_block(50_000_000, 10_000, 5)
w5 = torch.rand(50_000_000, dtype=torch.float64, device="cuda")
print(f"cfg3 histogram (50M x 10k) {best(lambda: c._sandwich_diag_dev(w5, None, None)):.4f}")
