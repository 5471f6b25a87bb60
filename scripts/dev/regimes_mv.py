"""SplitMatrix.matvec / transpose_matvec across regimes (2M rows): ms and effective GB/s
(operands read once).  usage: python scripts/dev/regimes_mv.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
N = 2_000_000
CASES = [
    ("cfg4 shape", dict()),
    ("float32", dict(dtype=torch.float32)),
    ("density 1 %", dict(density=0.01)),
    ("sparse 2048 cols @ 1.25 %", dict(k_sparse=2048, density=0.0125)),
    ("dense 50 cols (unaligned)", dict(k_dense=50)),
    ("dense 256 cols", dict(k_dense=256)),
    ("cats 10000 / 500", dict(cats=(10000, 500))),
    ("cats 5 x 20", dict(cats=(20, 20, 20, 20, 20))),
]
def best(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print(f"{'case':30s} {'matvec ms':>10s} {'GB/s':>7s} {'rmatvec ms':>11s} {'GB/s':>7s}")
for name, kw in CASES:
    X = synth.mixed_split(N, **kw)
    dt = kw.get("dtype", torch.float64)
    v = torch.rand(X.shape[1], dtype=dt, device="cuda")
    w = torch.rand(N, dtype=dt, device="cuda")
    b = synth.algorithmic_bytes(X) - X.shape[1] ** 2 * 8
    t1 = best(lambda: X.matvec(v))
    t2 = best(lambda: X.transpose_matvec(w))
    print(f"{name:30s} {t1:10.3f} {b / t1 / 1e6:7.0f} {t2:11.3f} {b / t2 / 1e6:7.0f}", flush=True)
    X = None
    torch.cuda.empty_cache()
