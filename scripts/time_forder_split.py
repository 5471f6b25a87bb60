"""cfg4 SplitMatrix with the dense block stored F-ordered: whole sandwich + per-op breakdown."""
import os, sys, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tabmat_amd as tm
from tabmat_amd import synth, _lib
n = 10_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
dm = X.matrices[0]
buf = dm._dev().buf
XF = tm.SplitMatrix([tm.DenseMatrix(buf.t().contiguous().t())] + list(X.matrices[1:]))
d = torch.rand(n, dtype=torch.float64, device="cuda")
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print(f"C-order dense block: sandwich {t(lambda: X._sandwich_dev(d, None, None)):7.3f} ms")
print(f"F-order dense block: sandwich {t(lambda: XF._sandwich_dev(d, None, None)):7.3f} ms")
_lib.call("tm_profile_enable", 1)
for name, M in (("C", X), ("F", XF)):
    dmat, smat = M.matrices[0], M.matrices[1]
    smat._cross_sandwich_dev(dmat, d, None, None, None)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms))
    print(f"  {name}: sparse x dense (K3) {ms.value:7.3f} ms")
