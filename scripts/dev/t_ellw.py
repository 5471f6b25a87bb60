import numpy as np, scipy.sparse as sps, torch, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd as tm
from tabmat_amd.ext import sparse as xs
rng = np.random.default_rng(0)
for dt in (np.float64, np.float32):
    for (n, m, nb, dens) in [(1000, 50, 128, 0.05), (777, 37, 100, 0.2), (5000, 300, 256, 0.03), (130, 16, 72, 0.5), (64, 3, 68, 1.0), (4099, 513, 132, 0.01)]:
        A = sps.random(n, m, dens, format="csc", random_state=1, dtype=np.float64).astype(dt)
        Bm = rng.standard_normal((n, nb)).astype(dt)
        d = rng.random(n).astype(dt); d[::7] = 0
        S = tm.SparseMatrix(A); Dn = tm.DenseMatrix(Bm)
        res = S._cross_sandwich(Dn, d, None)
        ref = (A.T.astype(np.float64) @ (d[:, None].astype(np.float64) * Bm.astype(np.float64)))
        err = np.abs(res - ref).max() / max(1, np.abs(ref).max())
        print(dt.__name__, n, m, nb, dens, "err", err, "wide", S._ell(wide=True).wide)
        assert err < (1e-12 if dt == np.float64 else 1e-4)
print("ok")
