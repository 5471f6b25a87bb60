"""-m gpu parity of K2b, the block-list sparse self sandwich (csrc/sparse_blocks.hip; reference:
ext/sparse.pyx:17-77) against the oracle and the chunked kernel."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

from _gpu_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(1, 128, 0.5), (777, 100, 0.1), (20_011, 300, 0.06), (9_000, 512, 0.05),
                                      (5_003, 640, 0.2), (3_000, 129, 0.6)])
def test_blocks_sandwich_vs_oracle(n, m, dens, dtype):
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    S = S.astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, max(1, n // 7))] = 0
    A = tm.SparseMatrix(S)._dev()
    got = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
    want = orc.sparse_sandwich(sps.csc_matrix(S).astype(np.float64), sps.csr_matrix(S).astype(np.float64),
                               d.astype(np.float64), None, None)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    assert rel_err(got, want) < tol
    assert np.array_equal(got, got.T)
    other = D.to_host(xs.sparse_sandwich_chunked(A, D.to_dev(d)))
    assert rel_err(got, other) < (1e-12 if dtype == np.float64 else 1e-4)


def test_blocks_path_is_taken_by_sparse_matrix_sandwich(monkeypatch):
    import tabmat_amd as tm
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(5)
    S = sps.random(30_000, 512, density=0.05, format="csc", random_state=rng)
    sm = tm.SparseMatrix(S)
    d = rng.random(30_000)
    called = []
    real = xs.sparse_sandwich_blocks
    monkeypatch.setattr(xs, "sparse_sandwich_blocks", lambda A, dd: (called.append(1), real(A, dd))[1])
    got = sm.sandwich(d)
    assert called
    assert rel_err(got, (S.T.multiply(d)).dot(S).toarray()) < 1e-10
    # a masked row restriction runs the same kernel
    rows = np.sort(rng.choice(30_000, 20_000, replace=False))
    Sr = S.tocsr()[rows]
    assert rel_err(sm.sandwich(d, rows), (Sr.T.multiply(d[rows])).dot(Sr).toarray()) < 1e-10


@pytest.mark.parametrize("waves", [2, 5, 8, 12])
def test_wave_knob_does_not_change_the_result(waves):
    """tm_tune_set("k2b_waves") launches fewer waves than the block table's FULL / HALF split was
    built for (16): the kernel rescales the split instead of skipping blocks."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd import _lib
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(3)
    S = sps.random(9_000, 300, density=0.08, format="csc", random_state=rng, dtype=np.float64)
    d = rng.random(9_000)
    A = tm.SparseMatrix(S)._dev()
    want = (S.T.multiply(d)).dot(S).toarray()
    _lib.call("tm_tune_set", b"k2b_waves", waves)
    try:
        got = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
    finally:
        _lib.call("tm_tune_set", b"k2b_waves", -2**63)
    assert rel_err(got, want) < 1e-10
