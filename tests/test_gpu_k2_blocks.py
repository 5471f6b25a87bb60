"""-m gpu parity of K2b, the block-list sparse self sandwich (csrc/sparse_blocks.hip; reference:
ext/sparse.pyx:17-77) against the oracle and the chunked kernel."""
import os
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
    # byte columns (the default, tm_sparse_sandwich_{blocks,chunked}_u8_*) against the int32 columns: the same pairs,
    # only the order of the LDS atomics differs
    assert xs.K2B_U8
    try:
        xs.K2B_U8 = False
        got32 = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
        other32 = D.to_host(xs.sparse_sandwich_chunked(A, D.to_dev(d)))
    finally:
        xs.K2B_U8 = True
    assert rel_err(got, got32) < (1e-12 if dtype == np.float64 else 1e-4)
    assert rel_err(other, other32) < (1e-12 if dtype == np.float64 else 1e-4)
    # round 6: the default list holds 12-byte descriptors (tm_sparse_sandwich_blocks_p12_*); the 16-byte list on byte
    # columns (tm_sparse_sandwich_blocks_u8_*: blocks of 2^24 rows or more, TABMAT_AMD_K2B_DESC12=0) gives the same
    from tabmat_amd.ext import _types as T

    assert int(A.pair_blocks()[0].shape[1]) == 3
    try:
        T.K2B_DESC12 = False
        got16 = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
        assert int(A.pair_blocks()[0].shape[1]) == 4
    finally:
        T.K2B_DESC12 = True
    assert rel_err(got, got16) < (1e-12 if dtype == np.float64 else 1e-4)


@pytest.mark.skipif(os.environ.get("TABMAT_AMD_DETERMINISTIC", "0") not in ("", "0"), reason="the fixed-order sparse self sandwich is selected instead")
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


@pytest.mark.parametrize("n,m,dens,n_wg,cyclic", [(20_011, 300, 0.06, 64, 128), (9_000, 512, 0.05, 1024, 4096),
                                                  (777, 100, 0.1, 16, 32), (50_000, 256, 0.03, 256, 1000)])
def test_round_robin_deal_of_row_ranges(n, m, dens, n_wg, cyclic):
    """Round 4: the blocks of a tile are dealt to its workgroups in round-robin row ranges (the default above 50M
    blocks, profiles/r4_k2b.txt); here forced on small matrices.  The table must cover every block exactly once,
    FULL blocks first inside a workgroup, rows inside [first row, last row], and the product must not change."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng, dtype=np.float64)
    d = rng.random(n)
    A = tm.SparseMatrix(S)._dev()
    ref_blocks, _, _ = A.pair_blocks()
    assert int(ref_blocks.shape[1]) == 3               # round 6: 12-byte descriptors below 2^24 rows
    ref_blocks = A.unpack_blocks(ref_blocks)
    n_blocks = int(ref_blocks.shape[0])
    ref_sorted = torch.unique(ref_blocks, dim=0)
    A._pb = None
    blocks, tab, max_nb = A.pair_blocks(n_wg=n_wg, cyclic=cyclic)
    blocks = A.unpack_blocks(blocks)
    assert int(blocks.shape[0]) == n_blocks and torch.equal(torch.unique(blocks, dim=0), ref_sorted)
    t = tab.cpu().numpy()
    b = blocks.cpu().numpy()
    covered = np.zeros(n_blocks, dtype=np.int64)
    for part, slot, lo, hi, fend, wf, r0, r1 in t:
        assert 0 <= slot < max_nb and lo < hi and lo <= fend <= hi and 0 <= wf <= 16
        covered[lo:hi] += 1
        rows = b[lo:hi, 2]
        assert rows.min() >= r0 and rows.max() <= r1
        na, nb = b[lo:hi, 3] & 0xff, (b[lo:hi, 3] >> 8) & 0xff
        full = (na > 4) & (nb > 4)
        assert full[:fend - lo].all() and not full[fend - lo:].any()
        assert (np.diff(rows[:fend - lo]) >= 0).all() and (np.diff(rows[fend - lo:]) >= 0).all()
    assert (covered == 1).all()
    got = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
    assert rel_err(got, (S.T.multiply(d)).dot(S).toarray()) < 1e-10
    assert np.array_equal(got, got.T)


def test_to_device_builds_the_block_list_up_front():
    """the block list's builder synchronises with the host (per-tile counts): to_device() builds it, so
    that the first sandwich -- possibly inside a HIP-graph capture -- finds it."""
    import tabmat_amd as tm

    rng = np.random.default_rng(8)
    S = sps.random(30_000, 512, density=0.05, format="csc", random_state=rng)
    sm = tm.SparseMatrix(S)
    assert getattr(sm._dev(), "_pb", None) is None
    sm.to_device()
    assert getattr(sm._dev(), "_pb", None) is not None


@pytest.mark.parametrize("deal", [dict(), dict(n_wg=96, cyclic=64)])
@pytest.mark.parametrize("kind", ["one_full_column", "dense_rows", "banded", "two_chunks_only"])
def test_blocks_sandwich_structured_patterns(kind, deal):
    """Patterns a uniform random matrix never shows: a column every row holds (its diagonal cell collects n pairs), a few
    completely filled rows (4 x 4 pieces of 8 per tile and row), a band (only tiles next to the diagonal hold blocks),
    entries in the first and last chunk only (empty tiles in between) -- with both deals of the blocks."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(len(kind))
    n, m = 6_000, 400
    S = sps.lil_matrix((n, m))
    if kind == "one_full_column":
        S[:, 130] = rng.standard_normal((n, 1))
        S[rng.integers(0, n, 300), rng.integers(0, m, 300)] = 0.5
    elif kind == "dense_rows":
        for r in rng.choice(n, 7, replace=False):
            S[r, :] = rng.standard_normal((1, m))
    elif kind == "banded":
        rows = np.repeat(np.arange(n), 5)
        cols = (np.repeat(np.arange(n) * m // n, 5) + np.tile(np.arange(5) * 9, n)) % m
        S = sps.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, m)).tolil()
    else:
        S[:, :20] = sps.random(n, 20, density=0.3, random_state=rng).tolil()
        S[:, m - 20:] = sps.random(n, 20, density=0.3, random_state=rng).tolil()
    S = sps.csc_matrix(S)
    S.sum_duplicates()
    d = rng.random(n)
    d[::11] = 0
    A = tm.SparseMatrix(S)._dev()
    if deal:
        A.pair_blocks(**deal)
    got = D.to_host(xs.sparse_sandwich_blocks(A, D.to_dev(d)))
    want = (S.T.multiply(d)).dot(S).toarray()
    assert rel_err(got, want) < 1e-10
    assert np.array_equal(got, got.T)
