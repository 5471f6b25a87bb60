"""Categoricals with many levels: the level-sorted cross-term kernels (csrc/cat_sorted.hip) against
the oracle, directly and through SplitMatrix.sandwich (which also exercises the grouping of the
small categoricals into fused passes)."""
import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from _gpu_util import to_tm_split

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("levels,drop,missing", [(5000, False, False), (700, True, True), (40_000, False, True)])
def test_sorted_kernels_match_oracle(levels, drop, missing, dtype):
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(levels)
    n, k, m = 30_011, 72, 300
    codes = rng.integers(0, levels, n).astype(np.int32)
    if missing:
        codes[rng.random(n) < 0.03] = -1
    cat = tm.CategoricalMatrix(codes, categories=np.arange(levels), drop_first=drop, dtype=dtype,
                               cat_missing_method="zero" if missing else "fail")
    B = rng.standard_normal((n, k)).astype(dtype)
    S = sps.random(n, m, density=0.04, format="csc", random_state=rng, dtype=np.float64).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 8)] = 0
    tol = 1e-10 if dtype == np.float64 else 2e-5
    ncol = cat.shape[1]
    got = D.to_host(xsplit.cat_dense_sandwich_sorted(cat._det_plan(), ncol, D.to_dev(d),
                                                     tm.DenseMatrix(B)._dev_c()))
    want = orc.sandwich_cat_dense(codes, ncol, d, B, None, None, drop_first=drop)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)
    got = D.to_host(xsplit.cat_sparse_sandwich_sorted(cat._det_plan(), ncol, D.to_dev(d),
                                                      tm.SparseMatrix(S)._dev()))
    want = orc.sandwich_cat_dense(codes, ncol, d, np.ascontiguousarray(S.toarray()), None, None, drop_first=drop)
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("cats", [(3000, 40), (20, 20, 20, 20, 20, 20), (600, 500, 7)])
def test_split_sandwich_with_many_levels(cats, order):
    from oracle import oracle as orc

    specs, idx = cs.mixed_specs(20_000, 64, 130, cats, seed=11, order=order)
    X = to_tm_split(specs, idx)
    rng = np.random.default_rng(4)
    d = rng.random(20_000)
    rows = np.sort(rng.choice(20_000, 12_000, replace=False))
    cols = np.arange(0, X.shape[1], 2)
    for r, c in ((None, None), (rows, None), (rows, cols)):
        got = X.sandwich(d, rows=r, cols=c)
        want = orc.split_sandwich([cs.to_oracle_block(s) for s in specs], idx, d, r, c)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()


def test_excluded_rows_may_hold_inf():
    import tabmat_amd as tm

    rng = np.random.default_rng(8)
    n = 10_000
    codes = rng.integers(0, 2000, n).astype(np.int32)
    B = rng.standard_normal((n, 32))
    d = rng.random(n)
    rows = np.sort(rng.choice(n, 7000, replace=False))
    excl = np.setdiff1d(np.arange(n), rows)
    cat = tm.CategoricalMatrix(codes, categories=np.arange(2000))
    want = cat._cross_sandwich(tm.DenseMatrix(B), d, rows)
    B2 = B.copy()
    B2[excl[:50], 3] = np.inf
    B2[excl[50:90], 7] = np.nan
    got = cat._cross_sandwich(tm.DenseMatrix(B2), d, rows)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens,k", [(30_011, 2048, 0.001, 72), (5000, 1500, 0.003, 128), (64, 1100, 0.002, 16)])
def test_column_sorted_sparse_dense(n, m, dens, k, dtype):
    """Sparse x dense for wide blocks with a few nonzeros per row: the column-sorted kernel
    (tm_csc_dense_sandwich_sorted_*) directly and through SparseMatrix._cross_sandwich."""
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    S = S.astype(dtype)
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 8)] = 0
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    want = orc.csr_dense_sandwich(sps.csr_matrix(S), B, d, None, None, None)
    got = D.to_host(xs.csc_dense_sandwich_sorted(sm._dev(), dm._dev_c(), D.to_dev(d)))
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)
    rows = np.sort(rng.choice(n, n // 2, replace=False))
    got = sm._cross_sandwich(dm, d, rows, np.arange(0, m, 2), np.arange(1, k, 3))
    want = orc.csr_dense_sandwich(sps.csr_matrix(S), B, d, rows, np.arange(0, m, 2), np.arange(1, k, 3))
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30)


def test_sorted_cat_sparse_wide_output():
    """More sparse columns than one LDS row of doubles holds (8192): several column passes."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(21)
    n, m, levels = 6000, 9100, 900
    codes = rng.integers(0, levels, n).astype(np.int32)
    S = sps.random(n, m, density=0.002, format="csr", random_state=rng, dtype=np.float64)
    d = rng.random(n)
    cat = tm.CategoricalMatrix(codes, categories=np.arange(levels))
    got = D.to_host(xsplit.cat_sparse_sandwich_sorted(cat._det_plan(), levels, D.to_dev(d),
                                                      tm.SparseMatrix(sps.csc_matrix(S))._dev()))
    oh = sps.csr_matrix((d, (codes, np.arange(n))), shape=(levels, n))
    want = (oh @ S).toarray()
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
