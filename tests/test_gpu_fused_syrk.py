"""The fused pass  sparse x dense cross term + dense self sandwich  (tm_csr_dense_sandwich_lg_syrk_f64:
one read of the dense block, the self sandwich on the matrix cores inside the gather kernel) against
the oracle.  The path is opt-in (TABMAT_AMD_FUSE_SYRK=1; measured slower than the two kernels, see
DESIGN.md) but must stay correct: ragged row counts, sparse widths that leave waves without columns,
zeros in d."""
import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from _gpu_util import to_tm_split

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [64, 20_011, 70_000])
@pytest.mark.parametrize("m", [512, 400, 257, 1000])
def test_fused_kernel_matches_oracle(n, m):
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=0.05, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    B = rng.standard_normal((n, 128))
    d = rng.random(n)
    d[rng.integers(0, n, n // 7)] = 0
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    lg = sm._lg()
    assert lg is not None and xs.lg_syrk_supported(lg, dm._dev_c())
    cross, selfb = xs.csr_dense_sandwich_lg_syrk(lg, dm._dev_c(), D.to_dev(d))
    want_x = orc.csr_dense_sandwich(sps.csr_matrix(S), B, d, None, None, None)
    want_s = orc.dense_sandwich(B, d, None, None)
    got_x, got_s = D.to_host(cross), D.to_host(selfb)
    assert np.abs(got_x - want_x).max() <= 1e-10 * np.abs(want_x).max()
    assert np.abs(got_s - want_s).max() <= 1e-10 * np.abs(want_s).max()
    assert np.array_equal(got_s, got_s.T) or np.abs(got_s - got_s.T).max() <= 1e-12 * np.abs(got_s).max()


def test_split_sandwich_through_fused_pass(monkeypatch):
    """SplitMatrix.sandwich with the fused pass switched on = the same result as with it off."""
    import tabmat_amd.split_matrix as smod
    from tabmat_amd import _device as D

    specs, idx = cs.mixed_specs(30_000, 128, 512, (256, 96, 32), seed=5)
    X = to_tm_split(specs, idx)
    d = np.random.default_rng(2).random(30_000)
    monkeypatch.setattr(smod, "FUSE_SYRK", False)
    want = X.sandwich(d)
    monkeypatch.setattr(smod, "FUSE_SYRK", True)
    got = X.sandwich(d)
    assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    got_dev = D.to_host(X.sandwich(D.to_dev(d), cols=np.arange(0, X.shape[1], 3)))
    monkeypatch.setattr(smod, "FUSE_SYRK", False)
    want_dev = X.sandwich(d, cols=np.arange(0, X.shape[1], 3))
    assert np.abs(got_dev - want_dev).max() <= 1e-11 * np.abs(want_dev).max()
