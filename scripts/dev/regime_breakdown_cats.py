"""Per-kernel times (rocprofv3-free: bench.kernel_breakdown) of the categorical-heavy regimes, 2M rows."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tabmat_amd import synth
N = 2_000_000
for name, kw in (("cats 10000 / 500", dict(cats=(10000, 500))), ("cats 12 x 30", dict(cats=(30,) * 12)),
                 ("cats 5 x 20", dict(cats=(20,) * 5))):
    X = synth.mixed_split(N, **kw)
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    for _ in range(2):
        X.sandwich(d)
    bd = bench.kernel_breakdown(X, d)
    print(name, " ".join(f"{k}={v:.2f}" for k, v in sorted(bd.items(), key=lambda kv: -kv[1]) if v > 0.03), flush=True)
    X = None
    torch.cuda.empty_cache()
