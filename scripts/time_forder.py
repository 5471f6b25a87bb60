"""Time dense-block products for C- and F-ordered storage at cfg4 size (10M x 128 float64)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tabmat_amd as tm
from tabmat_amd.ext._types import DenseDev
n, k = 10_000_000, 128
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
Xc = torch.randn((n, k), dtype=torch.float64, device="cuda", generator=g)
d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
v = torch.rand(k, dtype=torch.float64, device="cuda", generator=g)
for order in ("C", "F"):
    buf = Xc if order == "C" else Xc.t().contiguous().t()     # same values, column-major storage
    M = tm.DenseMatrix(buf)
    print(f"{order}-order: sandwich {t(lambda: M._sandwich_dev(d, None, None)):7.3f} ms   "
          f"matvec {t(lambda: M._matvec_dev(v, None, None, None, False)):7.3f} ms   "
          f"rmatvec {t(lambda: M._matvec_dev(d, None, None, None, True)):7.3f} ms")
