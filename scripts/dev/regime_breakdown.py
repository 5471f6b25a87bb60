"""Per-op main-kernel times (bench.kernel_breakdown) of regimes outside the tuned shape, 2M rows.
usage: python scripts/dev/regime_breakdown.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tabmat_amd import synth
N = 2_000_000
for name, kw in (("cfg4 shape", dict()), ("density 20 %", dict(density=0.20)),
                 ("sparse 2048 cols @ 1.25 %", dict(k_sparse=2048, density=0.0125)),
                 ("dense 256 cols", dict(k_dense=256)), ("sparse 4096 cols @ 0.2 %", dict(k_sparse=4096, density=0.002)),
                 ("sparse 8192 cols @ 0.05 %", dict(k_sparse=8192, density=0.0005))):
    X = synth.mixed_split(N, **kw)
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    for _ in range(2):
        X.sandwich(d)
    bd = bench.kernel_breakdown(X, d)
    print(name, " ".join(f"{k}={v:.2f}" for k, v in sorted(bd.items(), key=lambda kv: -kv[1]) if v > 0.05), flush=True)
    X = None
    torch.cuda.empty_cache()
