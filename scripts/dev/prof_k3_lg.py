"""Per-phase cycle counters of the lane-group K3 kernel (build with -DLG_PROF)."""
import os, sys, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = 10_000_000
dt = torch.float64
dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, 0.05, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
lg = sm._lg()
for _ in range(2):
    xs.csr_dense_sandwich_lg(lg, Bd, d, unc=2)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 16 * 8, dtype=np.int64)
raw.tm_lg_prof_fetch(buf.ctypes.data_as(C.c_void_p))
a = buf.reshape(256, 16, 8).astype(np.float64)
it = a[..., 4]
names = ["issue(top)", "compute", "vmcnt wait", "barrier wait"]
tot = a[..., :4].sum(axis=-1)
print("iterations per WG:", it.mean(), " clock64 ticks per iteration:", (tot / it).mean())
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {((a[..., i] / it).mean()):9.1f} ticks/iter  ({100 * a[..., i].sum() / tot.sum():5.1f} %)"
          f"   per-wave min {((a[..., i] / it).min()):8.1f} max {((a[..., i] / it).max()):8.1f}")
