"""One 40M-row shard of the cfg4 recipe (4x the benchmark size): fits one MI355X with every twin."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
n = int(os.environ.get("ROWS", 40_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
X.to_device()
torch.cuda.synchronize()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ms = t(lambda: X._sandwich_dev(d, None, None))
free, total = torch.cuda.mem_get_info()
print(f"{n} rows: sandwich {ms:.1f} ms = {13.52e9 * n / 10e6 / ms / 1e6:.0f} GB/s effective; HBM in use {(total - free) / 1e9:.0f} GB")
S = X._sandwich_dev(d, None, None)
print("symmetric:", bool(torch.allclose(S, S.T, rtol=1e-12, atol=0)), " finite:", bool(torch.isfinite(S).all()))
# round 6: ingest time / resident bytes at this size, and S u = X' (d * (X u)) from independent kernels
torch.cuda.empty_cache()
print(f"resident (live allocations) {torch.cuda.memory_allocated() / 1e9:.1f} GB for {13.51 * n / 10e6:.1f} GB of data")
u = torch.randn(X.shape[1], dtype=torch.float64, device="cuda")
lhs = S @ u
rhs = X.transpose_matvec(d * X.matvec(u))
print(f"|S u - X'(d * X u)| / |S u| = {float((lhs - rhs).abs().max() / lhs.abs().max()):.2e}")
