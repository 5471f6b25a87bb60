"""SplitMatrix.matvec / transpose_matvec at cfg4, 20 calls each (for a kernel trace: which launches a call makes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n)
v = torch.rand(X.shape[1], dtype=torch.float64, device="cuda")
w = torch.rand(n, dtype=torch.float64, device="cuda")
for name, f in (("matvec", lambda: X.matvec(v)), ("transpose_matvec", lambda: X.transpose_matvec(w))):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
