"""Entry-wise error (natural scale of the STANDARDIZED result, against long-double algebra) of
StandardizedMatrix.sandwich with and without the centred dense kernels, by the mean / std of the dense columns."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from scipy import sparse as sps
import tabmat_amd as tm
from tabmat_amd import dense_matrix as dm
from test_gpu_standardized_centered import _ld_sandwich, _standardized_ld
rng = np.random.default_rng(0)
n = 32768
for ratio in (0.0, 10.0, 400.0, 1e4):
    Xd = (ratio + rng.standard_normal((n, 72))) * 5.0
    mat = tm.SplitMatrix([tm.DenseMatrix(Xd), tm.SparseMatrix(sps.random(n, 24, 0.05, format="csc", random_state=rng)),
                          tm.CategoricalMatrix(rng.integers(0, 20, n))])
    std = mat.standardize(np.full(n, 1.0 / n), True, True)[0]
    d = rng.random(n)
    want = np.asarray(_ld_sandwich(_standardized_ld(std), d), dtype=np.float64)
    dg = np.sqrt(np.abs(np.diag(want))); den = np.outer(dg, dg)
    out = []
    for strict in (False, True):
        old = dm.set_strict_f64(strict)
        for cen in (True, False):
            s2 = tm.StandardizedMatrix(std.mat, std.shift, std.mult); s2.CENTER_DENSE = cen
            out.append(float((np.abs(s2.sandwich(d) - want) / den).max()))
        dm.set_strict_f64(old)
    print(f"mean/std {ratio:8.0f}:  int8 default  centred {out[0]:.1e}  reference formula {out[1]:.1e}   |   strict f64  centred {out[2]:.1e}  reference formula {out[3]:.1e}", flush=True)
