"""Ingest mirror of the reference's constructor (src/tabmat/constructor.py:29-212, 297-308;
tests/test_matrices.py::test_pandas_to_matrix, tests/test_split_matrix.py::test_init): host
logic on CPU, one product through the HIP path under -m gpu."""
import warnings

import numpy as np
import pandas as pd
import pytest
from scipy import sparse as sps

import tabmat_amd as tm


def _frame(n=300, seed=0):
    rng = np.random.default_rng(seed)
    return pd.DataFrame({
        "dense": rng.standard_normal(n),
        "sparse": np.where(rng.random(n) < 0.05, rng.random(n), 0.0),
        "cat": pd.Categorical(rng.integers(0, 7, n)),
        "small_cat": pd.Categorical(np.array(["u", "v", "w"])[rng.integers(0, 3, n)]),
        "flag": rng.random(n) < 0.5,
        "rare_flag": rng.random(n) < 0.03,
        "text": ["x"] * n,
    })


def _expected_dense(df, drop_first=False):
    """Column-by-column expansion in DataFrame order ('expand' placement)."""
    cols = []
    for name in df.columns:
        c = df[name]
        if str(c.dtype) == "category":
            codes = c.cat.codes.to_numpy()
            k = len(c.cat.categories)
            oh = np.zeros((len(c), k))
            oh[np.arange(len(c))[codes >= 0], codes[codes >= 0]] = 1.0
            cols.append(oh[:, int(drop_first):])
        elif c.dtype == bool or pd.api.types.is_numeric_dtype(c.dtype):
            cols.append(c.to_numpy().astype(float)[:, None])
    return np.hstack(cols)


@pytest.mark.parametrize("drop_first", [False, True])
def test_from_pandas_blocks_and_values(drop_first):
    df = _frame()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        X = tm.from_pandas(df, drop_first=drop_first)
    assert any("were ignored" in str(x.message) and "text" in str(x.message) for x in w)
    kinds = sorted(type(m).__name__ for m in X.matrices)
    assert kinds == ["CategoricalMatrix", "DenseMatrix", "SparseMatrix"]
    np.testing.assert_array_equal(X.toarray(), _expected_dense(df, drop_first))
    names = X.get_names("column")
    assert names[0] == "dense" and names[1] == "sparse"
    assert names[2] == ("cat[1]" if drop_first else "cat[0]")
    # below cat_threshold levels: one-hot columns, dense because each level has > 10 % of the rows
    assert any(str(n).startswith("small_cat[") for n in names)
    sp = [m for m in X.matrices if isinstance(m, tm.SparseMatrix)][0]
    assert set(sp.get_names("column")) == {"sparse", "rare_flag"}


def test_from_pandas_cat_position_end_and_object_as_cat():
    df = _frame()
    X = tm.from_pandas(df, cat_position="end", object_as_cat=True)
    names = X.get_names("column")
    plain = ["dense", "sparse", "flag", "rare_flag"]
    assert names[:4] == plain
    assert all("[" in str(n) for n in names[4:])
    assert "text[x]" in names                      # object column became a one-level categorical
    full = _expected_dense(df.assign(text=pd.Categorical(df["text"])))
    order = [list(df.columns).index(c) for c in plain]
    got = X.toarray()
    np.testing.assert_array_equal(got[:, :4], df[plain].to_numpy().astype(float))
    assert got.shape[1] == full.shape[1]
    np.testing.assert_array_equal(np.sort(got.sum(axis=0)), np.sort(full.sum(axis=0)))
    assert len(order) == 4


def test_from_pandas_single_block_and_errors():
    df = pd.DataFrame({"a": np.arange(5.0) + 1, "b": np.ones(5)})
    X = tm.from_pandas(df)
    assert isinstance(X, tm.DenseMatrix) and X.shape == (5, 2)
    with pytest.raises(ValueError, match="no valid column"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tm.from_pandas(pd.DataFrame({"t": ["a", "b"]}))
    miss = pd.DataFrame({"c": pd.Categorical(["a", None, "b", "a", "c", "d"])})
    with pytest.raises(ValueError, match="missing"):
        tm.from_pandas(miss)
    Z = tm.from_pandas(miss, cat_missing_method="zero")
    assert Z.toarray().sum() == 5
    Cv = tm.from_pandas(miss, cat_missing_method="convert")
    assert Cv.shape[1] == 5 and "c[(MISSING)]" in Cv.get_names("column")


def test_from_csc_split_by_density():
    rng = np.random.default_rng(3)
    n = 400
    cols = [rng.random(n) * (rng.random(n) < dens) for dens in (0.9, 0.02, 0.5, 0.05, 0.0, 0.11)]
    A = sps.csc_matrix(np.stack(cols, axis=1))
    X = tm.from_csc(A, threshold=0.1)
    dense = [m for m in X.matrices if isinstance(m, tm.DenseMatrix)][0]
    sparse = [m for m in X.matrices if isinstance(m, tm.SparseMatrix)][0]
    assert dense.shape[1] == 3 and sparse.shape[1] == 3
    np.testing.assert_array_equal(X.toarray(), A.toarray())
    with pytest.raises(TypeError):
        tm.from_csc(A.tocsr())
    with pytest.raises(ValueError, match="Threshold"):
        tm.from_csc(A, threshold=1.5)


@pytest.mark.gpu
def test_from_pandas_products_on_device():
    df = _frame(5000, seed=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X = tm.from_pandas(df)
    E = _expected_dense(df)
    rng = np.random.default_rng(5)
    d, v, w = rng.random(len(df)), rng.standard_normal(E.shape[1]), rng.standard_normal(len(df))
    ref = E.T @ (d[:, None] * E)
    assert np.abs(X.sandwich(d) - ref).max() / np.abs(ref).max() < 1e-10
    assert np.abs(X.matvec(v) - E @ v).max() < 1e-9
    assert np.abs(X.transpose_matvec(w) - E.T @ w).max() < 1e-9
