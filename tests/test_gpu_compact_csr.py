"""Round 6: after `to_device()` a sparse block of at most 65 536 columns keeps ONLY 16-bit CSR columns
(CsrDev.compact_indices; -1 GB at BASELINE configs[3]); everything that still wants int32 columns -- the generic
restricted kernels, row slicing, twin builders that run later, the host copy -- reads `.indices`, which widens into a
shared scratch.  The threshold (1M nonzeros) is lowered here so that small blocks take the path; every product is
compared with the oracle AFTER the compaction."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs
from _gpu_util import to_tm_split

pytestmark = pytest.mark.gpu


@pytest.fixture
def compact_everything(monkeypatch):
    from tabmat_amd.ext import sparse as xs

    monkeypatch.setattr(xs, "CSR_U16_MIN_NNZ", 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_products_after_the_int32_columns_are_gone(compact_everything, dtype):
    from oracle import oracle as orc
    from tabmat_amd.ext._types import release_index_scratch

    n = 30_000
    specs, idx = cs.mixed_specs(n, 72, 300, (11, 5), seed=31, dtype=dtype)
    X = to_tm_split(specs, idx, dtype).to_device()
    sm = X.matrices[1]
    A = sm._dev()
    assert A._ind32 is None and A._ind16 is not None
    specs64 = [(s[0], s[1].astype(np.float64)) + tuple(s[2:]) if s[0] != "cat" else s for s in specs]
    blocks = [cs.to_oracle_block(s) for s in specs64]          # (the oracle in float64 on the same values)
    rng = np.random.default_rng(4)
    d = rng.random(n).astype(dtype)
    v = rng.standard_normal(X.shape[1]).astype(dtype)
    rows = np.sort(rng.choice(n, n // 7, replace=False))
    cols = np.sort(rng.choice(X.shape[1], X.shape[1] // 3, replace=False))
    tol = 1e-10 if dtype == np.float64 else 3e-4

    def close(a, b):
        assert np.abs(np.asarray(a, dtype=np.float64) - b).max() <= tol * max(np.abs(b).max(), 1e-30)

    d64, v64 = d.astype(np.float64), v.astype(np.float64)
    close(X.sandwich(d), orc.split_sandwich(blocks, idx, d64))
    close(X.sandwich(d, rows=rows, cols=cols), orc.split_sandwich(blocks, idx, d64, rows, cols))     # generic kernels
    close(X.matvec(v), orc.split_matvec(blocks, idx, v64))
    close(X.matvec(v, cols=cols), orc.split_matvec(blocks, idx, v64, cols))
    close(X.transpose_matvec(d), orc.split_transpose_matvec(blocks, idx, d64))
    close(X.transpose_matvec(d, rows=rows, cols=cols), orc.split_transpose_matvec(blocks, idx, d64, rows, cols))
    # the sparse block alone: restricted self sandwich (int32 CSR kernels), a twin built AFTER the compaction
    S = specs[1][1].astype(np.float64)
    sc = np.sort(rng.choice(300, 80, replace=False)).astype(np.int32)
    close(sm.sandwich(d, rows=rows, cols=sc), orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d64, rows, sc))
    rws, vls, bstart, n_blocks, col_bptr = A.csc_blocks()
    assert int(rws.numel()) == S.nnz
    # row slicing on the device and the way back to the host
    sub = X[100:9000]
    close(sub.sandwich(d[100:9000]), orc.split_sandwich([cs.to_oracle_block(cs.take_rows(s, np.arange(100, 9000)))
                                                          for s in specs64], idx, d64[100:9000]))
    host = sm._host()
    assert (abs(host - sps.csc_matrix(specs[1][1])) > 0).nnz == 0
    release_index_scratch()
    close(sm.sandwich(d, rows=rows[:50], cols=sc), orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d64,
                                                                       rows[:50], sc))


def test_the_scratch_of_widened_columns_is_shared_and_bounded(compact_everything):
    """At most two widened arrays live at once, keyed weakly: a block that is gone releases its scratch."""
    import gc

    import tabmat_amd as tm
    from tabmat_amd.ext import _types as T

    rng = np.random.default_rng(0)
    mats = []
    for k in range(3):
        S = sps.random(5000, 90 + k, density=0.05, format="csc", random_state=rng)
        m = tm.SparseMatrix(S).to_device()
        assert m._dev()._ind32 is None
        mats.append((m, S))
    T.release_index_scratch()
    for m, S in mats:
        got = m._dev().indices.cpu().numpy()
        np.testing.assert_array_equal(got, S.tocsr().indices.astype(np.int32))
    assert len(T._WIDE) == 2
    a = mats[2][0]._dev().indices
    assert a.data_ptr() == mats[2][0]._dev().indices.data_ptr()          # cached, not widened again
    del a
    mats.clear()
    m = S = None
    gc.collect()
    T._WIDE[:] = [(o, t) for o, t in T._WIDE if o() is not None]
    assert len(T._WIDE) == 0
