import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth
def t(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter()-t0)*1e3:10.2f} ms", flush=True); return r
n, c = 5_000_000, 10_000
for zipf in (0.0, 1.1):
    X = t(f"gen zipf={zipf}", lambda: synth.cat_block(n, c, seed=2, zipf=zipf))
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    t("  sandwich diag", lambda: X._sandwich_diag_dev(ones, None, None))
    t("  sandwich diag (2nd)", lambda: X._sandwich_diag_dev(ones, None, None))
    t("  torch.bincount", lambda: torch.bincount(X._dev().to(torch.int64), minlength=c))
    t("  transpose_matvec", lambda: X.transpose_matvec(ones))
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    t("  index_add_", lambda: torch.zeros(c, dtype=torch.float64, device="cuda").index_add_(0, X._dev().to(torch.int64), d))
    v = torch.rand(c, dtype=torch.float64, device="cuda")
    t("  matvec", lambda: X.matvec(v))
