"""2-D operands of matvec / transpose_matvec (multi-right-hand-side kernels, csrc/multirhs.hip)
against dense algebra -- the reference computes these with scipy.sparse / NumPy BLAS
(sparse_matrix.py:252-268, dense_matrix.py:212-217; tests/test_matrices.py 2-D `other` cases)."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

pytestmark = pytest.mark.gpu


def _mat(kind, n, dtype, rng, order="C"):
    import tabmat_amd as tm

    if kind == "dense":
        X = rng.standard_normal((n, 11)).astype(dtype)
        return tm.DenseMatrix(X if order == "C" else np.asfortranarray(X))
    if kind == "sparse":
        return tm.SparseMatrix(sps.random(n, 23, 0.15, format="csc", random_state=5, dtype=np.float64).astype(dtype))
    return tm.SplitMatrix([tm.DenseMatrix(rng.standard_normal((n, 5)).astype(dtype)),
                           tm.SparseMatrix(sps.random(n, 17, 0.2, format="csc", random_state=7,
                                                      dtype=np.float64).astype(dtype))],
                          [np.array([0, 3, 4, 9, 20]), np.array([1, 2, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15,
                                                                  16, 17, 18, 19, 21])])


@pytest.mark.parametrize("kind,order", [("dense", "C"), ("dense", "F"), ("sparse", "C"), ("split", "C")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("K", [1, 3, 70])
@pytest.mark.parametrize("restrict", [False, True])
def test_two_dimensional_operands(kind, order, dtype, K, restrict):
    rng = np.random.default_rng(K)
    n = 1234
    mat = _mat(kind, n, dtype, rng, order)
    A = mat.toarray().astype(np.float64)
    p = mat.shape[1]
    rows = np.sort(rng.choice(n, 400, replace=False)) if restrict else None
    cols = np.sort(rng.choice(p, max(1, p // 2), replace=False)) if restrict else None
    tol = 1e-10 if dtype == np.float64 else 2e-4
    V = rng.standard_normal((p, K)).astype(dtype)
    Ac = A[:, cols] if cols is not None else A
    want = Ac @ (V[cols] if cols is not None else V).astype(np.float64)
    for vv in (V, torch.from_numpy(V).cuda()):
        got = mat.matvec(vv, cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        assert got.shape == (n, K)
        assert np.abs(got - want).max() / max(np.abs(want).max(), 1e-30) < tol
    W = rng.standard_normal((n, K)).astype(dtype)
    Ar = Ac[rows] if rows is not None else Ac
    want_t = Ar.T @ (W[rows] if rows is not None else W).astype(np.float64)
    for ww in (W, torch.from_numpy(W).cuda()):
        got = mat.transpose_matvec(ww, rows, cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        assert got.shape == want_t.shape
        assert np.abs(got - want_t).max() / max(np.abs(want_t).max(), 1e-30) < tol
