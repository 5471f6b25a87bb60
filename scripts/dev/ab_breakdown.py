"""Per-op main-kernel times of the cfg4 shape (N rows) + matvec / transpose_matvec block kernels, restricted to entry
points that round 4's library has too -- for a same-box A/B:  TABMAT_AMD_LIB_LAX=1 TABMAT_AMD_LIB=<old .so> python ..."""
import os, sys, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tabmat_amd import synth, _lib
from tabmat_amd.ext import split as xsplit, sparse as xs
xsplit.PACKED_CODES = False
xs.K2_PAIRS = "0"
N = int(os.environ.get("N", 4_000_000))
X = synth.mixed_split(N, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(N, dtype=torch.float64, device="cuda")
v = torch.rand(X.shape[1], dtype=torch.float64, device="cuda")
for _ in range(2):
    X.sandwich(d)
bd = bench.kernel_breakdown(X, d, reps=4)
_lib.call("tm_profile_enable", 1)
def t(f, k=4):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
dm, sm = X.matrices[0], X.matrices[1]
bd["dense.matvec"] = t(lambda: dm._matvec_dev(v[:128].contiguous(), None, None, None, False))
bd["dense.rmatvec"] = t(lambda: dm._matvec_dev(d, None, None, None, True))
bd["sparse.matvec"] = t(lambda: sm._matvec_dev(v[:512].contiguous(), None, None, None, False))
bd["sparse.rmatvec"] = t(lambda: sm._matvec_dev(d, None, None, None, True))
print(os.environ.get("TABMAT_AMD_LIB", "default"), " ".join(f"{k}={x:.3f}" for k, x in bd.items()), flush=True)
