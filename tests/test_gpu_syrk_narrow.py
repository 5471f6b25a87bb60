"""-m gpu parity of K1n, the dense syrk for blocks of at most 11 columns (csrc/syrk_narrow.hip; reference:
ext/dense_helpers-tmpl.cpp:266-311, C and F order): against the oracle for every width, both orders and
dtypes, row counts around the kernel's 64-row wave steps and its two-rows-per-turn loop."""
import numpy as np
import pytest

import _cases as cs
from _gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize("m", range(1, 12))
def test_narrow_syrk_vs_oracle(m, dtype, tol, order):
    import tabmat_amd as tm
    from oracle import oracle as orc

    for n in (1, 63, 64, 65, 511, 512, 513, 20_011):
        rng = np.random.default_rng(n * 13 + m)
        X = np.asarray(rng.standard_normal((n, m)).astype(dtype), order=order)
        d = (rng.random(n) - 0.2).astype(dtype)             # negative weights are legal
        got = tm.DenseMatrix(X).sandwich(d)
        want = orc.dense_sandwich(X.astype(np.float64), d.astype(np.float64), None, None)
        assert got.shape == (m, m) and np.array_equal(got, got.T)
        assert rel_err(got, want) < tol


def test_narrow_syrk_is_the_path_taken(monkeypatch):
    import tabmat_amd as tm
    from tabmat_amd import _lib

    rng = np.random.default_rng(1)
    X = rng.standard_normal((100_000, 10))
    d = rng.random(100_000)
    a = tm.DenseMatrix(X).sandwich(d)
    _lib.call("tm_tune_set", b"syrk_narrow", 0)
    try:
        b = tm.DenseMatrix(X).sandwich(d)
    finally:
        _lib.call("tm_tune_set", b"syrk_narrow", 1)
    assert rel_err(a, (X.T * d) @ X) < 1e-12 and rel_err(a, b) < 1e-12
