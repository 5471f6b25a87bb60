"""The reference's 'sparse_wide' benchmark design (40000 x 10000 at 1 %): SparseMatrix.sandwich by kernel
choice (K2_PAIRS auto / 0 / 1), parity against scipy."""
import time
import numpy as np
import torch
from scipy import sparse as sps
import tabmat_amd as tm
from tabmat_amd.ext import sparse as xs

rng = np.random.default_rng(0)
for n, m, dens in ((40_000, 10_000, 0.01), (400_000, 10_000, 0.001), (40_000, 16_000, 0.005)):
    S = sps.random(n, m, density=dens, format="csc", random_state=rng)
    d = rng.random(n)
    ref = (S.T.multiply(d)).dot(S)
    for mode in ("0", "1", "auto"):
        xs.K2_PAIRS = mode
        mat = tm.SparseMatrix(S)
        dd = torch.from_numpy(d).cuda()
        out = mat.sandwich(dd) if hasattr(mat, "sandwich") else None
        for _ in range(2):
            out = mat.sandwich(dd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = mat.sandwich(dd)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        o = out.cpu().numpy() if torch.is_tensor(out) else np.asarray(out)
        err = abs(o - ref.toarray()).max() / abs(ref).max()
        print(f"n={n} m={m} dens={dens} K2_PAIRS={mode}: {ms:.3f} ms  err {err:.2e}  pays={xs.pairs_sandwich_pays(mat._dev()) if mode=='auto' else ''}", flush=True)
