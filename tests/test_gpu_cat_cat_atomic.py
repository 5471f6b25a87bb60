"""-m gpu: categorical x categorical tables too large for a few LDS tiles (reference: ext/split.pyx:83-111,
cat_split_helpers-tmpl.cpp:44-94) -- the global-atomic form (tm_cat_cat_sandwich_atomic_*) that evenly
filled tables take, against the oracle, and the host-side guard that keeps skewed tables off it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc

    return orc


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("drop", [(False, False), (True, False), (True, True)])
def test_large_even_table_takes_the_atomic_form(dtype, drop, monkeypatch):
    import tabmat_amd as tm
    from tabmat_amd import _lib

    seen = []
    real = _lib.call
    monkeypatch.setattr("tabmat_amd.ext.split.call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    rng = np.random.default_rng(3)
    n, ki, kj = 200_000, 1000, 900
    ci, cj = rng.integers(0, ki, n), rng.integers(0, kj, n)
    A = tm.CategoricalMatrix(ci, drop_first=drop[0], dtype=dtype)
    B = tm.CategoricalMatrix(cj, drop_first=drop[1], dtype=dtype)
    d = rng.random(n).astype(dtype)
    got = A._cross_sandwich(B, d)
    assert any(s.startswith("tm_cat_cat_sandwich_atomic_") for s in seen)
    assert A.shape[1] == ki - drop[0] and B.shape[1] == kj - drop[1]
    want = _orc().sandwich_cat_cat(ci.astype(np.int32), cj.astype(np.int32), A.shape[1], B.shape[1],
                                   d.astype(np.float64), None, drop[0], drop[1])
    tol = 1e-12 if dtype == np.float64 else 1e-5
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1.0)


def test_skewed_table_keeps_the_lds_form(monkeypatch):
    import tabmat_amd as tm
    from tabmat_amd import _lib

    seen = []
    real = _lib.call
    monkeypatch.setattr("tabmat_amd.ext.split.call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    rng = np.random.default_rng(4)
    n, k = 200_000, 1000
    ci = np.where(rng.random(n) < 0.5, 7, rng.integers(0, k, n))          # half of the rows in one level
    cj = np.where(rng.random(n) < 0.5, 3, rng.integers(0, k, n))
    A, B = tm.CategoricalMatrix(ci), tm.CategoricalMatrix(cj)
    d = rng.random(n)
    got = A._cross_sandwich(B, d)
    assert not any(s.startswith("tm_cat_cat_sandwich_atomic_") for s in seen)
    want = _orc().sandwich_cat_cat(ci.astype(np.int32), cj.astype(np.int32), k, k, d, None, False, False)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
