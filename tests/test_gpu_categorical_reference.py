"""The reference's tests/test_categorical_matrix.py and tests/test_big_categorical_matrix.py:10-90
restated for tabmat_amd (same inputs, seeds and assertions): drop_first x missing x
cat_missing_method in {fail, zero, convert} through matvec / transpose_matvec / tocsr / multiply /
indexing, and the 797 586-row / 58 059-level block with READ-ONLY index buffers.  The
polars / pyarrow / narwhals extraction cases are DataFrame-library ingest (out of
scope) -- the pandas and list variants are kept."""
import re

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _cat_vec(missing):
    rng = np.random.default_rng(0)
    vec = rng.choice([0, 1, 2, np.inf, -np.inf], size=10)
    if missing:
        vec[vec == 1] = np.nan
    return vec


GRID = pytest.mark.parametrize(
    "drop_first,missing,cat_missing_method",
    [(df, mi, me) for df in (True, False) for mi in (True, False) for me in ("fail", "zero", "convert")])


def _make(cat_vec, drop_first, missing, method):
    """The matrix, or None after checking the ValueError of the `fail` method."""
    import tabmat_amd as tm

    if missing and method == "fail":
        with pytest.raises(ValueError, match="Categorical data can't have missing values"):
            tm.CategoricalMatrix(cat_vec, drop_first=drop_first, cat_missing_method=method)
        return None
    return tm.CategoricalMatrix(cat_vec, drop_first=drop_first, cat_missing_method=method)


def _dummies(cat_vec, drop_first, missing, method):
    return pd.get_dummies(cat_vec, drop_first=drop_first, dtype="uint8",
                          dummy_na=(method == "convert" and missing))


@GRID
def test_recover_orig(drop_first, missing, cat_missing_method):
    cat_vec = _cat_vec(missing)
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is not None:
        np.testing.assert_equal(mat.recover_orig(), cat_vec)


@pytest.mark.parametrize("vec_dtype", [np.float64, np.float32, np.int64, np.int32])
@GRID
def test_csr_matvec_categorical(vec_dtype, drop_first, missing, cat_missing_method):
    cat_vec = _cat_vec(missing)
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is None:
        return
    ref = _dummies(cat_vec, drop_first, missing, cat_missing_method)
    vec = np.random.choice(np.arange(4, dtype=vec_dtype), ref.shape[1])
    np.testing.assert_allclose(mat.matvec(vec), mat.toarray().dot(vec))


@GRID
def test_tocsr(drop_first, missing, cat_missing_method):
    cat_vec = _cat_vec(missing)
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is not None:
        np.testing.assert_allclose(mat.tocsr().toarray(),
                                   _dummies(cat_vec, drop_first, missing, cat_missing_method))


@GRID
def test_transpose_matvec(drop_first, missing, cat_missing_method):
    cat_vec = _cat_vec(missing)
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is None:
        return
    other = np.random.random(mat.shape[0])
    want = _dummies(cat_vec, drop_first, missing, cat_missing_method).T.dot(other)
    np.testing.assert_allclose(mat.transpose_matvec(other), want)


@GRID
def test_multiply(drop_first, missing, cat_missing_method):
    cat_vec = _cat_vec(missing)
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is None:
        return
    other = np.arange(len(cat_vec))[:, None]
    want = _dummies(cat_vec, drop_first, missing, cat_missing_method) * other
    np.testing.assert_allclose(mat.multiply(other).toarray(), want)


@pytest.mark.parametrize("mi_element", [np.nan, None])
def test_nulls(mi_element):
    import tabmat_amd as tm

    with pytest.raises(ValueError, match="Categorical data can't have missing values"):
        tm.CategoricalMatrix([0, mi_element, 1])


@pytest.mark.parametrize("cat_missing_name", ["(MISSING)", "__None__", "[NULL]"])
def test_cat_missing_name(cat_missing_name):
    import tabmat_amd as tm

    vec = [None, "(MISSING)", "__None__", "a", "b"]
    if cat_missing_name in vec:
        with pytest.raises(ValueError,
                           match=re.escape(f"Missing category {cat_missing_name} already exists.")):
            tm.CategoricalMatrix(vec, cat_missing_method="convert", cat_missing_name=cat_missing_name)
    else:
        cat = tm.CategoricalMatrix(vec, cat_missing_method="convert", cat_missing_name=cat_missing_name)
        assert set(cat.categories) == set(vec) - {None} | {cat_missing_name}


@GRID
def test_categorical_indexing(drop_first, missing, cat_missing_method):
    cat_vec = [0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 3] if not missing else \
        [0, None, 2, 0, None, 2, 0, None, 2, 3, 3]
    mat = _make(cat_vec, drop_first, missing, cat_missing_method)
    if mat is None:
        return
    want = pd.get_dummies(cat_vec, drop_first=drop_first,
                          dummy_na=cat_missing_method == "convert" and missing).to_numpy()[:, [0, 1]]
    np.testing.assert_allclose(mat[:, [0, 1]].toarray(), want)


@pytest.mark.parametrize("input_type", ["pandas.Categorical", "pandas", "list"])
def test_extract_codes_and_categories(input_type):
    import tabmat_amd as tm

    cat_vec = pd.Series(["a", "b", "c", pd.NA, "b", "a", "d"], dtype="category")
    if input_type == "pandas.Categorical":
        cat_vec = pd.Categorical(cat_vec)
    elif input_type == "list":
        cat_vec = cat_vec.astype("object")
    mat = tm.CategoricalMatrix(cat_vec, cat_missing_method="zero")
    np.testing.assert_array_equal(mat.indices, np.array([0, 1, 2, -1, 1, 0, 3]))
    np.testing.assert_array_equal(mat.categories, np.array(["a", "b", "c", "d"]))


def test_shape_of_empty():
    import tabmat_amd as tm

    assert tm.CategoricalMatrix([], drop_first=True).shape == (0, 0)


# ---- tests/test_big_categorical_matrix.py:10-90 ------------------------------------------------
N_BIG, K_BIG = 797_586, 58_059


def _big(n, n_categories, **kw):
    import tabmat_amd as tm

    categories = [f"cat[{i}]" for i in range(n_categories)]
    indices = np.linspace(0, n_categories - 1, n).round().astype(int)
    cat_vec = pd.Series(pd.Categorical.from_codes(indices, categories=categories))
    mat = tm.CategoricalMatrix(cat_vec, **kw)
    mat._host_codes.flags.writeable = False          # the reference's index buffer is read-only here
    return mat, indices


def test_transpose_matvec_does_not_crash():
    mat, indices = _big(N_BIG, K_BIG)
    res = mat.transpose_matvec(np.ones(N_BIG))
    assert res is not None
    np.testing.assert_array_equal(res, np.bincount(indices, minlength=K_BIG))      # exact counts


def test_sandwich_cat_cat_does_not_crash():
    for na, nb in [(K_BIG, 2725), (2725, K_BIG)]:
        A, ia = _big(N_BIG, na)
        B, ib = _big(N_BIG, nb)
        w = np.ones(N_BIG) / N_BIG
        res = A._cross_categorical(B, w, np.arange(N_BIG), np.arange(na), np.arange(nb))
        assert res is not None and res.shape == (na, nb)
        want = np.zeros((na, nb))
        np.add.at(want, (ia, ib), w)
        np.testing.assert_allclose(res, want, rtol=1e-10, atol=1e-15)


@pytest.mark.parametrize("drop_first", [False, True])
def test_cross_dense_does_not_crash(drop_first):
    import tabmat_amd as tm

    mat, indices = _big(N_BIG, K_BIG, drop_first=drop_first)
    assert not mat.indices.flags.writeable
    dense = tm.DenseMatrix(np.ones((N_BIG, 10)))
    w = np.ones(N_BIG) / N_BIG
    res = mat._cross_sandwich(dense, w, np.arange(N_BIG), np.arange(mat.shape[1]), np.arange(10))
    assert res is not None
    want = np.bincount(indices, weights=w, minlength=K_BIG)[int(drop_first):]
    np.testing.assert_allclose(res, np.repeat(want[:, None], 10, axis=1), rtol=1e-10)


# ---- deterministic K4a (ext/cat_split_helpers-tmpl.cpp:33-38, CHANGELOG.rst) -------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("drop_first,missing", [(False, False), (True, True)])
def test_deterministic_transpose_matvec_is_bitwise_reproducible(dtype, drop_first, missing, monkeypatch):
    import tabmat_amd as tm
    import tabmat_amd.categorical_matrix as cm

    rng = np.random.default_rng(3)
    n, k = 300_000, 37
    codes = rng.zipf(1.3, n) % k                      # skewed: long runs of one category
    if missing:
        codes[rng.random(n) < 0.05] = -1
    v = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(dtype)   # order-sensitive sums
    mat = tm.CategoricalMatrix(codes, categories=np.arange(k), drop_first=drop_first, dtype=dtype,
                               cat_missing_method="zero" if missing else "fail")
    want = np.zeros(k)
    np.add.at(want, codes[codes >= 0], v[codes >= 0].astype(np.float64))
    want = want[int(drop_first):]
    monkeypatch.setattr(cm, "DETERMINISTIC", True)
    runs = [mat.transpose_matvec(v) for _ in range(4)]
    for r in runs[1:]:
        np.testing.assert_array_equal(r, runs[0])               # bit for bit
    tol = 1e-10 if dtype == np.float64 else 1e-3
    assert np.abs(runs[0] - want).max() / np.abs(want).max() < tol
    # row / column restrictions and the sandwich diagonal go through the same kernel
    rows = np.sort(rng.choice(n, n // 3, replace=False))
    cols = np.sort(rng.choice(mat.shape[1], 11, replace=False))
    sel = np.zeros(n, bool)
    sel[rows] = True
    w2 = np.zeros(k)
    ok = (codes >= 0) & sel
    np.add.at(w2, codes[ok], v[ok].astype(np.float64))
    w2 = w2[int(drop_first):][cols]
    got = mat.transpose_matvec(v, rows, cols)
    assert np.abs(got - w2).max() / np.abs(w2).max() < tol
    d1 = mat.sandwich(np.abs(v)).diagonal()
    d2 = mat.sandwich(np.abs(v)).diagonal()
    np.testing.assert_array_equal(d1, d2)
    # counts stay exact
    monkeypatch.setattr(cm, "DETERMINISTIC", True)
    ones = np.ones(n, dtype=dtype)
    cnt = np.bincount(codes[codes >= 0], minlength=k)[int(drop_first):]
    np.testing.assert_array_equal(mat.transpose_matvec(ones), cnt)
