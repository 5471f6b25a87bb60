"""Pair twin (K2 v4): the static stream of the tiled sparse self sandwich.  CPU: the twin decodes
back to the matrix (slots, padding, overflow entries, group pointers).  GPU: kernel vs oracle over
densities that exercise empty lists, full base slots and overflow on both sides, ragged n, zero
weights."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

from tabmat_amd.ext._types import CsrDev, PairTwin


def _csr_cpu(S, dtype):
    S = sps.csr_matrix(S).astype(dtype)
    S.sort_indices()
    return CsrDev(torch.from_numpy(S.data.copy()), torch.from_numpy(S.indices.astype(np.int32)),
                  torch.from_numpy(S.indptr.astype(np.int64)), S.shape[0], S.shape[1])


def _decode(tw: PairTwin):
    n, m = tw.n, tw.m
    nch, G = (m + 127) // 128, (n + 7) // 8
    bv, bk = tw.bv.numpy().reshape(nch, G, 8, 8), tw.bk.numpy().reshape(nch, G, 8, 8)
    optr = tw.optr.numpy().reshape(nch, G + 1)
    ov, ok = tw.ov.numpy(), tw.ok.numpy()
    out = np.zeros((n, m))
    for c in range(nch):
        for g in range(G):
            for r in range(8):
                seen_pad = False
                last = -1
                for t in range(8):
                    col = bk[c, g, r, t]
                    if col < 0:
                        seen_pad = True
                        assert bv[c, g, r, t] == 0
                        continue
                    assert not seen_pad and col > last        # compact, ascending
                    last = col
                    out[8 * g + r, 128 * c + col] = bv[c, g, r, t]
            a, b = optr[c, g], optr[c, g + 1]
            assert 0 <= b - a <= 64
            for e in range(a, b):
                r, col = ok[e] >> 7, ok[e] & 127
                assert (bk[c, g, r] >= 0).all() and col > bk[c, g, r, 7]   # only after a full base
                out[8 * g + r, 128 * c + col] = ov[e]
    assert optr[-1, G] == tw.n_ov
    return out


@pytest.mark.parametrize("n,m,dens", [(100, 130, 0.05), (64, 128, 0.10), (37, 300, 0.08), (9, 5, 0.5)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_twin_decodes_to_the_matrix(n, m, dens, dtype):
    rng = np.random.default_rng(n * m)
    S = sps.random(n, m, density=dens, format="csr", random_state=rng, dtype=np.float64)
    S.data += 1.5
    tw = PairTwin.from_csr(_csr_cpu(S, dtype), max_overflow=1.0)
    assert tw is not None
    np.testing.assert_array_equal(_decode(tw), S.astype(dtype).toarray())


def test_twin_refuses_what_does_not_fit():
    rng = np.random.default_rng(3)
    dense = sps.random(3000, 1024, density=0.6, format="csr", random_state=rng)
    assert PairTwin.from_csr(_csr_cpu(dense, np.float64)) is None           # overflow fraction
    thin = sps.random(200_000, 4000, density=0.00002, format="csr", random_state=rng)
    assert PairTwin.from_csr(_csr_cpu(thin, np.float64)) is None            # padding


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(20_011, 512, 0.05), (5000, 300, 0.08), (70_001, 129, 0.02),
                                      (4096, 1000, 0.06), (17, 128, 0.3), (30_000, 64, 0.2)])
def test_pair_kernel_matches_oracle(n, m, dens, dtype):
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    S = S.astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 6)] = 0
    sm = tm.SparseMatrix(S)
    tw = PairTwin.from_csr(sm._dev(), max_overflow=1.0)
    assert tw is not None
    got = D.to_host(xs.sparse_sandwich_pair(tw, D.to_dev(d)))
    want = orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d, None, None)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    # the chunk-major form computes the same
    ref = D.to_host(xs.sparse_sandwich_chunked(sm._dev(), D.to_dev(d)))
    assert np.abs(got - ref).max() <= tol * np.abs(want).max()


@pytest.mark.gpu
def test_excluded_rows_may_hold_inf():
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(5)
    n, m = 6000, 256
    S = sps.random(n, m, density=0.08, format="csr", random_state=rng, dtype=np.float64)
    d = rng.random(n)
    bad = rng.choice(n, 500, replace=False)
    d[bad] = 0
    want = (S.T @ sps.diags(d) @ S).toarray()
    S2 = S.copy().tolil()
    for r in bad[:200]:
        cols = S2.rows[r]
        if cols:
            S2[r, cols[0]] = np.inf
            S2[r, cols[-1]] = np.nan
    S2 = sps.csr_matrix(S2)
    tw = PairTwin.from_csr(tm.SparseMatrix(S2)._dev(), max_overflow=1.0)
    got = D.to_host(xs.sparse_sandwich_pair(tw, D.to_dev(d)))
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
