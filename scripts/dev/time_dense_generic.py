"""Dense self-sandwich kernels through the plain C entry point tm_dense_sandwich_f64 (no int8): K1c, the generic MFMA
syrk with a row list, F order without the twin, a 256-column block (panels + rectangle) -- for A/B of two library builds."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import _lib
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext._types import DenseDev
n = 4_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.randn((n, 128), dtype=torch.float64, device="cuda", generator=g)
d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
rows = torch.arange(0, n, 2, dtype=torch.int32, device="cuda")
_lib.call("tm_profile_enable", 1)
def t(f, k=4):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
blkC = DenseDev(X, n, 128, 0)
XF = X.t().contiguous()
blkF = DenseDev(XF, n, 128, 1)
X256 = torch.randn((n // 2, 256), dtype=torch.float64, device="cuda", generator=g)
blk256 = DenseDev(X256, n // 2, 256, 0)
d2 = d[: n // 2].contiguous()
print(f"{os.environ.get('TABMAT_AMD_LIB', 'default'):40s} K1c {t(lambda: xd.dense_sandwich(blkC, d, None, None)):.3f}  "
      f"generic rows {t(lambda: xd.dense_sandwich(blkC, d, rows, None)):.3f}  F order {t(lambda: xd.dense_sandwich(blkF, d, None, None)):.3f}  "
      f"256 cols {t(lambda: xd.dense_sandwich(blk256, d2, None, None)):.3f} ms")
