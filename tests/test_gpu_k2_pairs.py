"""K2e, the pair-stream sparse self sandwich for wide blocks (csrc/sparse_pairs.hip, tm_sparse_sandwich_pairs_*):
parity with the oracle's restatement of ext/sparse.pyx:17-77 entry by entry at the natural scale, the shapes the
kernel's bookkeeping can get wrong (ragged last chunk, empty chunks, one-entry lists, lists longer than the
prefetched head, d == 0 rows holding inf, row segments shorter than a range), both dtypes, rows / cols through the
public method, and the dispatch."""
import os

import numpy as np
import pytest
import torch
from scipy import sparse as sps

from _gpu_util import nat_err

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc

    return orc


def _pairs(S, d, dtype=np.float64):
    import tabmat_amd as tm
    from tabmat_amd.ext import sparse as xs

    sm = tm.SparseMatrix(S.astype(dtype))
    dd = torch.from_numpy(d.astype(dtype)).cuda()
    # both record forms: packed {value, row << 7 | column in chunk} (12 / 8 bytes, the default) and 16-byte records
    outs = {}
    keep = xs.K2_PAIRS_PACKED
    try:
        for pk in (False, True):
            xs.K2_PAIRS_PACKED = pk
            outs[pk] = xs.sparse_sandwich_pairs(sm._dev(), dd).cpu().numpy()
    finally:
        xs.K2_PAIRS_PACKED = keep
    scale = max(1.0, float(np.abs(outs[False]).max())) if np.isfinite(outs[False]).all() else 1.0
    assert np.allclose(outs[True], outs[False], rtol=0, atol=(1e-12 if dtype == np.float64 else 1e-5) * scale, equal_nan=True)
    return outs[True]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(20_000, 2048, 0.0125), (9_001, 1500, 0.02), (30_000, 4096, 0.003),
                                      (5_000, 130, 0.1), (4_097, 512, 0.05), (700, 40, 0.5), (2_049, 8192, 0.001),
                                      (64, 300, 0.02), (3, 129, 0.9), (1_500, 10_000, 0.01), (300, 16_384, 0.002)])
def test_pairs_kernel_matches_the_oracle(dtype, n, m, dens):
    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng)
    d = rng.random(n)
    d[::5] = 0.0
    got = _pairs(S, d, dtype)
    S64 = S.astype(dtype).astype(np.float64)
    ref = _orc().sparse_sandwich(sps.csc_matrix(S64), sps.csr_matrix(S64), d.astype(dtype).astype(np.float64), None, None)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert got.shape == (m, m) and nat_err(got, ref) < tol
    assert np.array_equal(got, got.T)


def test_pairs_kernel_structured_patterns():
    """A full column, a banded block, rows with long lists in one chunk (beyond the prefetched head), empty
    column chunks in the middle, a block whose only entries sit in the last (ragged) chunk."""
    rng = np.random.default_rng(5)
    n, m = 6000, 1100
    A = sps.lil_matrix((n, m))
    A[:, 7] = rng.random((n, 1))                                  # full column
    for r in range(0, n, 3):                                      # band
        c = (r * 7) % (m - 3)
        A[r, c:c + 3] = rng.random(3)
    A[100:160, 256:300] = rng.random((60, 44))                    # 44 entries of a row in ONE chunk
    A[:, 384:512] = 0                                             # an empty chunk
    A[5000:5010, 1090:1100] = rng.random((10, 10))                # ragged last chunk
    S = A.tocsc()
    d = rng.random(n)
    got = _pairs(S, d)
    ref = (S.T.multiply(d)).dot(S).toarray()
    assert nat_err(got, ref) < 1e-10
    E = sps.lil_matrix((n, m))
    E[17, 1099] = 2.0
    E[4000, 1025] = 3.0
    got = _pairs(E.tocsc(), np.ones(n))
    want = np.zeros((m, m))
    want[1099, 1099], want[1025, 1025] = 4.0, 9.0
    assert np.array_equal(got, want)


def test_pairs_kernel_excluded_rows_may_hold_inf():
    rng = np.random.default_rng(6)
    n, m = 8000, 1300
    S = sps.random(n, m, density=0.01, format="csr", random_state=rng)
    d = rng.random(n)
    bad = rng.choice(n, 50, replace=False)
    d[bad] = 0.0
    S = S.tolil()
    for r in bad[:25]:
        S[r, int(rng.integers(0, m))] = np.inf
    S = S.tocsc()
    got = _pairs(S, d)
    clean = S.copy().tolil()
    for r in bad:
        clean[r, :] = 0
    clean = clean.tocsc()
    ref = (clean.T.multiply(d)).dot(clean).toarray()
    assert np.isfinite(got).all() and nat_err(got, ref) < 1e-10


@pytest.mark.skipif(os.environ.get("TABMAT_AMD_DETERMINISTIC", "0") not in ("", "0"),
                    reason="the fixed-order sparse self sandwich is selected instead")
def test_public_sandwich_takes_the_pairs_kernel_for_wide_blocks(monkeypatch):
    """SparseMatrix.sandwich with the pair-stream form switched on: unrestricted, a long row list (masked d), a
    short one (row-list kernels), a column selection -- all against dense algebra."""
    import tabmat_amd as tm
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(7)
    n, m = 25_000, 2048
    S = sps.random(n, m, density=0.0125, format="csc", random_state=rng)
    d = rng.random(n)
    monkeypatch.setattr(xs, "K2_PAIRS", "1")
    sm = tm.SparseMatrix(S)
    calls = []
    orig = xs.sparse_sandwich_pairs
    monkeypatch.setattr(xs, "sparse_sandwich_pairs", lambda A, dd: (calls.append(1), orig(A, dd))[1])
    ref = (S.T.multiply(d)).dot(S).toarray()
    assert nat_err(sm.sandwich(d), ref) < 1e-10 and len(calls) == 1
    rows = np.sort(rng.choice(n, n // 2, replace=False))
    Sr = S.tocsr()[rows]
    ref_r = (Sr.T.multiply(d[rows])).dot(Sr).toarray()
    assert nat_err(sm.sandwich(d, rows=rows), ref_r) < 1e-10 and len(calls) == 2
    few = np.sort(rng.choice(n, n // 20, replace=False))
    Sf = S.tocsr()[few]
    assert nat_err(sm.sandwich(d, rows=few), (Sf.T.multiply(d[few])).dot(Sf).toarray()) < 1e-10
    assert len(calls) == 2                                           # short list: not through the masked pass
    cols = np.sort(rng.choice(m, 900, replace=False))
    assert nat_err(sm.sandwich(d, cols=cols), ref[np.ix_(cols, cols)]) < 1e-10
