"""-m gpu parity tests, part 1: the reference's own small test cases
(/root/reference/tests/test_matrices.py, test_split_matrix.py, test_real_matrix.py) run through
the tabmat_amd classes -> C ABI -> HIP kernels, and compared with BOTH the CPU oracle and the
dense algebra the reference tests use."""
import os

import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from _gpu_util import nat_err, sub, to_tm_block, to_tm_split

pytestmark = pytest.mark.gpu

UNSCALED = cs.unscaled_specs()
IDS = [n for n, _ in UNSCALED]
PAIRS = [(a, b) for a in IDS for b in IDS
         if not (a.startswith("dense") and b.startswith("dense"))
         and not (a.startswith("sparse") and b.startswith("sparse"))
         and "drop" not in a and "drop" not in b]


def _orc():
    from oracle import oracle as orc

    return orc


@pytest.mark.parametrize("pair", PAIRS, ids=[f"{a}-{b}" for a, b in PAIRS])
@pytest.mark.parametrize("rows", [None, [2], np.arange(2)])
@pytest.mark.parametrize("L_cols", [None, [1], np.arange(1)])
@pytest.mark.parametrize("R_cols", [None, [1], np.arange(1)])
def test_cross_sandwich(pair, rows, L_cols, R_cols):
    """tests/test_matrices.py:348-392."""
    si, sj = dict(UNSCALED)[pair[0]], dict(UNSCALED)[pair[1]]
    mi, mj = to_tm_block(si), to_tm_block(sj)
    d = np.random.random(3)
    ai = sub(cs.spec_toarray(si), rows, L_cols)
    aj = sub(cs.spec_toarray(sj), rows, R_cols)
    dd = d if rows is None else d[np.asarray(rows)]
    res = mi._cross_sandwich(mj, d, rows, L_cols, R_cols)
    np.testing.assert_almost_equal(res, ai.T @ np.diag(dd) @ aj)
    orc = _orc()
    ref = orc.cross_sandwich(cs.to_oracle_block(si), cs.to_oracle_block(sj), d, rows, L_cols, R_cols)
    np.testing.assert_allclose(res, ref, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", IDS)
@pytest.mark.parametrize("vec_type", [lambda x: x, np.array])
@pytest.mark.parametrize("rows", [None, [], [1], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [0], np.arange(1)])
def test_self_sandwich(name, vec_type, rows, cols):
    """tests/test_matrices.py:395-413."""
    spec = dict(UNSCALED)[name]
    mat = to_tm_block(spec)
    vec_list = [3, 0.1, 1]
    res = mat.sandwich(vec_type(vec_list), rows, cols)
    if sps.issparse(res):
        res = res.toarray()
    m = sub(cs.spec_toarray(spec), rows, cols)
    vv = np.array(vec_list)
    vv = vv if rows is None else vv[np.asarray(rows, dtype=int)]
    np.testing.assert_allclose(res, m.T @ np.diag(vv) @ m)


@pytest.mark.parametrize("rows", [None, [], [0], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [0], np.arange(1), [1, 5, 6, 12]])
def test_split_sandwich(rows, cols):
    """tests/test_matrices.py:416-432."""
    specs, idx = cs.complex_split_specs()
    mat = to_tm_split(specs, idx)
    d = np.random.random(3)
    result = mat.sandwich(d, rows=rows, cols=cols)
    M = sub(mat.toarray(), rows, cols)
    dd = d if rows is None else d[np.asarray(rows, dtype=int)]
    np.testing.assert_almost_equal(result, M.T @ np.diag(dd) @ M)
    orc = _orc()
    ref = orc.split_sandwich([cs.to_oracle_block(s) for s in specs], idx, d, rows, cols)
    np.testing.assert_allclose(result, ref, rtol=1e-12, atol=1e-14)
    assert result.dtype == np.float64


def _all_mats():
    specs, idx = cs.complex_split_specs()
    return [(n, s, None) for n, s in UNSCALED] + [("split", specs, idx)]


def _build(entry):
    name, s, idx = entry
    if idx is None:
        return to_tm_block(s), cs.spec_toarray(s)
    m = to_tm_split(s, idx)
    return m, m.toarray()


@pytest.mark.parametrize("entry", _all_mats(), ids=[e[0] for e in _all_mats()])
@pytest.mark.parametrize("cols", [None, [], [1], np.array([1])])
@pytest.mark.parametrize("other_shape", [[], [1], [2]])
def test_matvec(entry, cols, other_shape):
    """tests/test_matrices.py:255-310."""
    import tabmat_amd as tm

    mat, A = _build(entry)
    other = np.random.random([mat.shape[1]] + other_shape)
    has_cat = isinstance(mat, tm.CategoricalMatrix) or (
        isinstance(mat, tm.SplitMatrix) and any(isinstance(m, tm.CategoricalMatrix) for m in mat.matrices))
    if has_cat and other.ndim > 1:
        with pytest.raises(NotImplementedError, match="only implemented for 1d"):
            mat.matvec(other, cols)
        return
    res = mat.matvec(other, cols)
    if cols is None:
        expected = A.dot(other)
    else:
        c = np.asarray(cols, dtype=int)
        expected = A[:, c].dot(other[c])
    np.testing.assert_allclose(res, expected, atol=1e-14)
    assert isinstance(res, np.ndarray)


@pytest.mark.parametrize("entry", _all_mats(), ids=[e[0] for e in _all_mats()])
@pytest.mark.parametrize("rows", [None, [], [2], np.arange(2)])
@pytest.mark.parametrize("cols", [None, [], [1], np.arange(1)])
def test_transpose_matvec(entry, rows, cols):
    """tests/test_matrices.py:313-345."""
    mat, A = _build(entry)
    other = np.array([3.0, -0.1, 0])
    res = mat.transpose_matvec(other, rows, cols)
    vv = other if rows is None else other[np.asarray(rows, dtype=int)]
    np.testing.assert_allclose(res, sub(A, rows, cols).T.dot(vv), atol=1e-14)
    assert isinstance(res, np.ndarray)


@pytest.mark.parametrize("entry", _all_mats(), ids=[e[0] for e in _all_mats()])
@pytest.mark.parametrize("cols", [None, [], [1], np.array([1])])
def test_matvec_out_parameter(entry, cols):
    """tests/test_matrices.py:129-143: out is modified in place and returned."""
    mat, _ = _build(entry)
    out = np.random.rand(mat.shape[0])
    out_copy = out.copy()
    v = np.random.rand(mat.shape[1])
    out2 = mat.matvec(v, cols=cols, out=out)
    assert out2 is out
    np.testing.assert_almost_equal(out, out_copy + mat.matvec(v, cols=cols))
    with pytest.raises(ValueError, match="first dimension of 'out' must be"):
        mat.matvec(v, cols, np.zeros(mat.shape[0] + 1))


@pytest.mark.parametrize("entry", _all_mats(), ids=[e[0] for e in _all_mats()])
@pytest.mark.parametrize("cols", [None, [], [1], np.array([0, 1])])
@pytest.mark.parametrize("rows", [None, [], [1], np.array([0, 2])])
def test_transpose_matvec_out_parameter(entry, cols, rows):
    """tests/test_matrices.py:146-171."""
    mat, A = _build(entry)
    out = np.random.rand(mat.shape[1])
    out_copy = out.copy()
    v = np.random.rand(mat.shape[0])
    out2 = mat.transpose_matvec(v, rows=rows, cols=cols, out=out)
    assert out2 is out
    col_idx = np.arange(mat.shape[1]) if cols is None else np.asarray(cols, dtype=int)
    row_idx = np.arange(mat.shape[0]) if rows is None else np.asarray(rows, dtype=int)
    part = A[row_idx, :][:, col_idx].T.dot(v[row_idx])
    correct = out_copy
    correct[col_idx] += part
    np.testing.assert_almost_equal(out, correct)
    with pytest.raises(ValueError, match="dimension of 'out' must be"):
        mat.transpose_matvec(v, rows, cols, np.zeros(mat.shape[1] + 1))


@pytest.mark.parametrize("entry", _all_mats(), ids=[e[0] for e in _all_mats()])
def test_error_conventions(entry):
    """tests/test_matrices.py:174-216: ValueError on misaligned shapes, TypeError on dtype."""
    mat, _ = _build(entry)
    n, m = mat.shape
    mat.sandwich(np.ones(n))
    for bad in (n - 1, n + 1):
        with pytest.raises(ValueError, match="not aligned"):
            mat.sandwich(np.ones(bad))
        with pytest.raises(ValueError):
            mat.transpose_matvec(np.ones(bad))
    for bad in (m - 1, m + 1):
        with pytest.raises(ValueError):
            mat.matvec(np.ones(bad))
    with pytest.raises(TypeError, match="same dtype"):
        mat.astype(np.float64).sandwich(np.ones(n, dtype=np.float32))
    with pytest.raises(TypeError, match="same dtype"):
        mat.astype(np.float32).sandwich(np.ones(n, dtype=np.float64))


@pytest.mark.parametrize("missing", [False, True])
@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("cols", [None, [0], [1, 2, 3], [1, 5]])
def test_split_with_cat_sandwich(missing, idx64, cols):
    """tests/test_split_matrix.py:170-189."""
    specs, idx = cs.split_with_cat_specs(missing, idx64)
    mat = to_tm_split(specs, idx)
    M = mat.toarray()
    for _ in range(3):
        v = np.random.rand(M.shape[0])
        y1 = mat.sandwich(v, cols=cols)
        Ml = M if cols is None else M[:, cols]
        np.testing.assert_allclose(y1, (Ml.T * v[None, :]) @ Ml, atol=1e-12)


@pytest.mark.parametrize("missing", [False, True], ids=["no_missing", "missing"])
def test_many_types(missing):
    """tests/test_split_matrix.py:249-288."""
    for i in range(10):
        specs, idx = cs.random_split_specs(
            seed=(1 if i == 0 else None), n_rows=1 + np.random.randint(130),
            n_cols_per=1 + np.random.randint(10), missing=missing)
        mat = to_tm_split(specs, idx)
        M = mat.toarray()
        d = np.random.random(M.shape[0])
        np.testing.assert_allclose(mat.sandwich(d), (M.T * d[None, :]) @ M, atol=1e-12)
        np.testing.assert_almost_equal(mat.transpose_matvec(d), M.T.dot(d))
        v = np.random.random(M.shape[1])
        np.testing.assert_almost_equal(mat.matvec(v), M.dot(v))


@pytest.mark.parametrize("idx64", [False, True])
@pytest.mark.parametrize("name", ["real_matrix_blocks.npz", "real_matrix_blocks_sparse.npz",
                                  "real_matrix_blocks_mixed.npz"])
def test_real_matrix_golden(name, idx64):
    """The reference's only data fixture (tests/test_real_matrix.py), standardized as there; the
    _sparse / _mixed variants split it with other thresholds so that a sparse block exists
    (tests/golden/make_real_matrix_fixture.py), loaded with int32 and int64 CSC indices."""
    import tabmat_amd as tm

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    blocks, idx = [], []
    for b, kind in enumerate(z["kinds"]):
        idx.append(z[f"b{b}_indices"])
        if kind == "cat":
            blocks.append(tm.CategoricalMatrix(z[f"b{b}_codes"],
                                               categories=np.arange(int(z[f"b{b}_ncat"]))))
        elif kind == "dense":
            blocks.append(tm.DenseMatrix(z[f"b{b}_array"]))
        else:
            S = sps.csc_matrix(z[f"b{b}_array"])
            if idx64:
                S = sps.csc_matrix((S.data, S.indices.astype(np.int64), S.indptr.astype(np.int64)),
                                   shape=S.shape)
            blocks.append(tm.SparseMatrix(S))
    X = tm.SplitMatrix(blocks, idx)
    np.testing.assert_array_equal(X.toarray(), z["design"])
    np.testing.assert_allclose(X.sandwich(z["d"]), z["sandwich"], rtol=1e-12, atol=1e-12)
    assert nat_err(X.sandwich(z["d"]), z["sandwich"]) < 1e-10
    assert nat_err(X.sandwich(z["d"], z["rows"], z["cols"]), z["sandwich_rows_cols"]) < 1e-10
    np.testing.assert_allclose(X.matvec(z["v"]), z["matvec"], rtol=1e-12)
    np.testing.assert_allclose(X.transpose_matvec(z["w"]), z["transpose_matvec"], rtol=1e-12,
                               atol=1e-12)
    np.testing.assert_allclose(X.sandwich(z["d"], z["rows"], z["cols"]), z["sandwich_rows_cols"],
                               rtol=1e-12, atol=1e-12)
    # tests/test_real_matrix.py:17-33: standardized split vs dense sandwich, 12 decimals
    n = X.shape[0]
    wts = np.ones(n) / n
    X_std = X.standardize(wts, True, True)[0]
    r = np.random.rand(n)
    dense = X_std.toarray()
    np.testing.assert_almost_equal(X_std.sandwich(r), (dense.T * r) @ dense, 12)
