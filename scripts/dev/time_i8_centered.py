"""K1e with and without the centre (tm_dense_sandwich_i8_centered_f64) at cfg4's dense block, and the whole
StandardizedMatrix.sandwich of the cfg4 recipe with / without centred dense terms."""
import os, sys, ctypes as C, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
import tabmat_amd as tm
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm = X.matrices[0]
c = torch.randn(128, dtype=torch.float64, device="cuda") * 0.1
_lib.call("tm_profile_enable", 1)
def t(f, k=5):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
print(f"K1e plain    {t(lambda: dm._sandwich_dev(d, None, None)):.3f} ms")
print(f"K1e centred  {t(lambda: dm._sandwich_dev(d, None, None, center=c)):.3f} ms")
print(f"K1e xtd      {t(lambda: dm._sandwich_xtd_dev(d)):.3f} ms")
print(f"K1e xtd cen  {t(lambda: dm._sandwich_xtd_dev(d, c)):.3f} ms")
_lib.call("tm_profile_enable", 0)
w = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
shift = np.random.default_rng(0).standard_normal(X.shape[1]); mult = np.random.default_rng(1).uniform(0.5, 1.5, X.shape[1])
for cen in (True, False):
    std = tm.StandardizedMatrix(X, shift, mult); std.CENTER_DENSE = cen
    for _ in range(2): std.sandwich(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): std.sandwich(d)
    torch.cuda.synchronize(); print(f"std.sandwich centred={cen}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
for _ in range(2): X.sandwich(d)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): X.sandwich(d)
torch.cuda.synchronize(); print(f"X.sandwich: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
