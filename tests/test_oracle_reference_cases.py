"""Pins the CPU oracle (oracle/) against the reference's own known-answer
tests, restated case by case, and against its only data fixture.  Every
expected value is recomputed from dense numpy algebra exactly as the reference
tests do (/root/reference/tests/test_matrices.py, test_fast_sandwich.py,
test_split_matrix.py, test_real_matrix.py).  CPU-only (-m "not gpu").
"""
import os

import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from oracle import oracle as orc

UNSCALED = cs.unscaled_specs()
IDS = [n for n, _ in UNSCALED]


def _sub(A, rows, cols):
    if rows is not None:
        A = A[np.asarray(rows, dtype=int), :]
    if cols is not None:
        A = A[:, np.asarray(cols, dtype=int)]
    return A


# ---- tests/test_fast_sandwich.py:12-31 ------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fast_sandwich_sparse(dtype):
    np.random.seed(123)
    for _ in range(10):
        nrows, ncols = np.random.randint(200, size=2)
        A = cs.simulate_matrix(shape=(nrows, ncols), seed=None, dtype=dtype).tocsc()
        d = np.random.rand(A.shape[0]).astype(dtype)
        true = (A.T.multiply(d)).dot(A).toarray()
        out = orc.sparse_sandwich(A, A.tocsr(), d, None, None)
        np.testing.assert_allclose(true, out, atol=np.sqrt(np.finfo(dtype).eps))


# ---- tests/test_fast_sandwich.py:51-98 -------------------------------------
def test_fast_sandwich_dense():
    np.random.seed(7)
    for _ in range(5):
        A = cs.simulate_matrix(shape=np.random.randint(1, 1000, size=2), seed=None)
        d = np.random.rand(A.shape[0])
        d[np.random.choice(np.arange(A.shape[0]), size=min(10, A.shape[0]), replace=False)] = 0.0
        for cols in (
            np.arange(A.shape[1], dtype=np.int32),
            np.random.choice(np.arange(A.shape[1]), size=np.random.randint(A.shape[1]),
                             replace=False).astype(np.int32),
        ):
            Asub = A[:, cols]
            true = (Asub.T.multiply(d)).dot(Asub).toarray()
            nonzero = np.where(np.abs(d) > 1e-14)[0].astype(np.int32)
            for arr in (np.asfortranarray(A.toarray()), np.ascontiguousarray(A.toarray())):
                out = orc.dense_sandwich(arr, d, nonzero, cols)
                np.testing.assert_allclose(true, out, atol=np.sqrt(np.finfo(np.float64).eps))


# ---- tests/test_split_matrix.py:147-166 ------------------------------------
@pytest.mark.parametrize("Acols", [np.arange(2, dtype=np.int32), np.array([1], dtype=np.int32)])
@pytest.mark.parametrize("Bcols", [np.arange(4, dtype=np.int32), np.array([1], dtype=np.int32),
                                   np.array([1, 3], dtype=np.int32)])
@pytest.mark.parametrize("order", ["C", "F"])
def test_sandwich_sparse_dense(Acols, Bcols, order):
    N = 100
    X = np.zeros((N, 4), order=order)
    X[:, 0] = 1.0
    X[:10, 1] = 0.5
    X[-20:, 2] = 0.25
    X[:, 3] = 2.0
    np.random.seed(0)
    d = np.random.random((N,))
    A = sps.random(N, 2).tocsr()
    result = orc.csr_dense_sandwich(A, X, d, None, Acols, Bcols)
    expected = A.T.toarray()[Acols, :] @ np.diag(d) @ X[:, Bcols]
    np.testing.assert_allclose(result, expected)


# ---- tests/test_matrices.py:348-392 (test_cross_sandwich, all 23 pairs) -----
PAIRS = [(a, b) for a in IDS for b in IDS
         if not (a.startswith("dense") and b.startswith("dense"))
         and not (a.startswith("sparse") and b.startswith("sparse"))
         and "drop" not in a and "drop" not in b]


@pytest.mark.parametrize("pair", PAIRS, ids=[f"{a}-{b}" for a, b in PAIRS])
@pytest.mark.parametrize("rows", [None, [2], np.arange(2)])
@pytest.mark.parametrize("L_cols", [None, [1], np.arange(1)])
@pytest.mark.parametrize("R_cols", [None, [1], np.arange(1)])
def test_cross_sandwich(pair, rows, L_cols, R_cols):
    si, sj = dict(UNSCALED)[pair[0]], dict(UNSCALED)[pair[1]]
    bi, bj = cs.to_oracle_block(si), cs.to_oracle_block(sj)
    d = np.random.random(3)
    mi = _sub(cs.spec_toarray(si), rows, L_cols)
    mj = _sub(cs.spec_toarray(sj), rows, R_cols)
    dd = d if rows is None else d[np.asarray(rows)]
    expected = mi.T @ np.diag(dd) @ mj
    res = orc.cross_sandwich(bi, bj, d, rows, L_cols, R_cols)
    np.testing.assert_almost_equal(res, expected)


# ---- tests/test_matrices.py:395-413 (test_self_sandwich) --------------------
@pytest.mark.parametrize("name", IDS)
@pytest.mark.parametrize("rows", [None, [], [1], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [0], np.arange(1)])
def test_self_sandwich(name, rows, cols):
    spec = dict(UNSCALED)[name]
    b = cs.to_oracle_block(spec)
    vec = np.array([3, 0.1, 1])
    res = orc.block_sandwich(b, vec, rows, cols)
    if b.kind == "cat":
        res = np.diag(res)
    m = _sub(cs.spec_toarray(spec), rows, cols)
    vv = vec if rows is None else vec[np.asarray(rows, dtype=int)]
    np.testing.assert_allclose(res, m.T @ np.diag(vv) @ m)


# ---- tests/test_matrices.py:416-432 (test_split_sandwich) -------------------
@pytest.mark.parametrize("rows", [None, [], [0], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [0], np.arange(1), [1, 5, 6, 12]])
def test_split_sandwich(rows, cols):
    specs, idx = cs.complex_split_specs()
    blocks = [cs.to_oracle_block(s) for s in specs]
    d = np.random.random(3)
    result = orc.split_sandwich(blocks, idx, d, rows, cols)
    M = _sub(orc.split_toarray(blocks, idx), rows, cols)
    dd = d if rows is None else d[np.asarray(rows, dtype=int)]
    np.testing.assert_almost_equal(result, M.T @ np.diag(dd) @ M)


# ---- tests/test_matrices.py:255-345 (matvec / transpose_matvec) -------------
@pytest.mark.parametrize("cols", [None, [], [1], np.array([1])])
def test_split_matvec(cols):
    specs, idx = cs.complex_split_specs()
    blocks = [cs.to_oracle_block(s) for s in specs]
    p = sum(len(i) for i in idx)
    v = np.random.random(p)
    res = orc.split_matvec(blocks, idx, v, cols)
    M = orc.split_toarray(blocks, idx)
    if cols is not None:
        c = np.asarray(cols, dtype=int)
        expected = M[:, c] @ v[c]
    else:
        expected = M @ v
    np.testing.assert_allclose(res, expected)


@pytest.mark.parametrize("rows", [None, [], [2], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [1], np.arange(1), [0, 3, 8]])
def test_split_transpose_matvec(rows, cols):
    specs, idx = cs.complex_split_specs()
    blocks = [cs.to_oracle_block(s) for s in specs]
    other = np.array([3.0, -0.1, 0])
    res = orc.split_transpose_matvec(blocks, idx, other, rows, cols)
    M = _sub(orc.split_toarray(blocks, idx), rows, cols)
    vv = other if rows is None else other[np.asarray(rows, dtype=int)]
    np.testing.assert_allclose(res, M.T @ vv)


# ---- tests/test_split_matrix.py:170-189 (test_sandwich with col subsets) ----
@pytest.mark.parametrize("missing", [False, True])
@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("cols", [None, [0], [1, 2, 3], [1, 5]])
def test_split_with_cat_sandwich(missing, idx64, cols):
    specs, idx = cs.split_with_cat_specs(missing, idx64)
    blocks = [cs.to_oracle_block(s) for s in specs]
    M = orc.split_toarray(blocks, idx)
    for _ in range(10):
        v = np.random.rand(M.shape[0])
        y1 = orc.split_sandwich(blocks, idx, v, None, cols)
        Ml = M if cols is None else M[:, cols]
        np.testing.assert_allclose(y1, (Ml.T * v[None, :]) @ Ml, atol=1e-12)


# ---- tests/test_split_matrix.py:249-288 (many random types) -----------------
@pytest.mark.parametrize("missing", [False, True], ids=["no_missing", "missing"])
def test_many_types(missing):
    for i in range(10):
        specs, idx = cs.random_split_specs(
            seed=(1 if i == 0 else None), n_rows=1 + np.random.randint(130),
            n_cols_per=1 + np.random.randint(10), missing=missing)
        blocks = [cs.to_oracle_block(s) for s in specs]
        M = orc.split_toarray(blocks, idx)
        d = np.random.random(M.shape[0])
        np.testing.assert_allclose(orc.split_sandwich(blocks, idx, d), (M.T * d[None, :]) @ M,
                                   atol=1e-12)
        np.testing.assert_almost_equal(orc.split_transpose_matvec(blocks, idx, d), M.T.dot(d))
        v = np.random.random(M.shape[1])
        np.testing.assert_almost_equal(orc.split_matvec(blocks, idx, v), M.dot(v))


# ---- categorical counts are exact -------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_categorical_counts_bit_exact(dtype):
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 1000, 200_000).astype(np.int32)
    ones = np.ones(len(codes), dtype=dtype)
    diag = orc.sandwich_categorical(codes, ones, None, 1000)
    assert np.array_equal(diag, np.bincount(codes, minlength=1000).astype(dtype))
    out = np.zeros(1000, dtype=dtype)
    orc.cat_transpose_matvec(codes, ones, 1000, None, None, out)
    assert np.array_equal(out, np.bincount(codes, minlength=1000).astype(dtype))


# ---- the reference's only data fixture (tests/test_real_matrix.py) ----------
REAL_FIXTURES = ["real_matrix_blocks.npz", "real_matrix_blocks_sparse.npz", "real_matrix_blocks_mixed.npz"]


def _load_real(name="real_matrix_blocks.npz", idx64=False):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    blocks, idx = [], []
    for b, kind in enumerate(z["kinds"]):
        idx.append(z[f"b{b}_indices"])
        if kind == "cat":
            blocks.append(orc.Cat(z[f"b{b}_codes"], int(z[f"b{b}_ncat"])))
        elif kind == "dense":
            blocks.append(orc.Dense(z[f"b{b}_array"]))
        else:
            S = sps.csc_matrix(z[f"b{b}_array"])
            if idx64:       # ext/sparse.pyx:13-15 win_integral: int32 or int64 index arrays
                S = sps.csc_matrix((S.data, S.indices.astype(np.int64), S.indptr.astype(np.int64)),
                                   shape=S.shape)
            blocks.append(orc.Sparse(S))
    return z, blocks, idx


@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("name", REAL_FIXTURES)
def test_real_matrix_golden(name, idx64):
    z, blocks, idx = _load_real(name, idx64)
    np.testing.assert_array_equal(orc.split_toarray(blocks, idx), z["design"])
    np.testing.assert_allclose(orc.split_sandwich(blocks, idx, z["d"]), z["sandwich"],
                               rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(orc.split_matvec(blocks, idx, z["v"]), z["matvec"], rtol=1e-12)
    np.testing.assert_allclose(orc.split_transpose_matvec(blocks, idx, z["w"]),
                               z["transpose_matvec"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(
        orc.split_sandwich(blocks, idx, z["d"], z["rows"], z["cols"]),
        z["sandwich_rows_cols"], rtol=1e-12, atol=1e-12)
