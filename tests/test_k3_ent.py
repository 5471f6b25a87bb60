"""Entry twin of a sparse block (tabmat_amd/ext/_types.py::SlabEnt) and the run-time-indexed sparse x dense
kernel on it (csrc/sparse_ent.hip, round 4; reference: ext/sparse.pyx:211-260 csr_dense_sandwich ->
ext/sparse_helpers-tmpl.cpp:23-146).

CPU part: the twin builder is plain torch, so its output is decoded here exactly the way the kernel walks it
(groups, batches of 16 slots, meta = slab tag | row in slab | column, bstart) and compared with the matrix.  GPU part: the
kernel through the C ABI against the oracle's csr_dense_sandwich -- float64 within 1e-10 (observed 1e-15),
float32 within 2e-5 of the float64 oracle."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

from tabmat_amd.ext._types import CsrDev, SlabEnt

R, C, U = 64, 16, 16


def _csr_cpu(S, dtype):
    S = sps.csr_matrix(S).astype(dtype)
    S.sort_indices()
    return CsrDev(torch.from_numpy(S.data.copy()), torch.from_numpy(S.indices.astype(np.int32)),
                  torch.from_numpy(S.indptr.astype(np.int64)), S.shape[0], S.shape[1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,density", [(300, 40, 0.05), (64, 16, 0.5), (129, 33, 0.2), (1000, 512, 0.02),
                                         (5, 3, 1.0), (200, 20, 0.0), (64 * 150 + 7, 20, 0.0002)])
def test_twin_decodes_to_the_matrix(n, m, density, dtype):
    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=density, format="csr", random_state=rng, dtype=np.float64)
    S.data += 0.5          # no explicit zeros
    tw = SlabEnt.from_csr(_csr_cpu(S, dtype))
    G = tw.mk // C
    nS = (n + R - 1) // R
    assert tw.bstart.shape == (G, nS + 1)
    vals, meta, bst = tw.vals.numpy(), tw.meta.numpy().view(np.uint16), tw.bstart.numpy().view(np.uint32)
    assert vals.shape[0] == meta.shape[0] == int(bst[-1, -1]) * U + SlabEnt.SLACK
    assert not vals[int(bst[-1, -1]) * U:].any() and not meta[int(bst[-1, -1]) * U:].any()    # the slack
    dense = np.zeros((tw.mk, n))
    prev_end = 0
    for g in range(G):
        assert bst[g, 0] == prev_end                      # groups follow one another
        for s in range(nS):
            b0, b1 = int(bst[g, s]), int(bst[g, s + 1])
            assert b1 >= b0
            real = 0
            for q in range(b0 * U, b1 * U):
                # round 6: 16-bit words {slab & 63, row in slab, column in group}
                tag, r6, col = int(meta[q]) >> 10, (int(meta[q]) >> 4) & 63, int(meta[q]) & 15
                assert tag == (s & 63)
                row = s * R + r6
                assert row < n                             # every slot (padding too) names a row of ITS slab
                if vals[q] != 0:
                    assert dense[g * C + col, row] == 0
                    dense[g * C + col, row] = vals[q]
                    real += 1
            # whole batches; an EMPTY block holds none -- except the continuity batch at every 32nd slab (a matrix
            # without any entry has no stream at all)
            assert (b1 - b0) == (real + U - 1) // U + int(real == 0 and s % 32 == 0 and S.nnz > 0)
        prev_end = int(bst[g, nS])
    np.testing.assert_array_equal(dense[tw.inv.numpy()].T, S.toarray().astype(dtype))


def test_twin_refuses_what_the_kernel_cannot_index():
    S = sps.random(4000, 2048, density=0.0005, format="csr", random_state=np.random.default_rng(0))
    assert SlabEnt.from_csr(_csr_cpu(S, np.float64), max_pad=8.0) is not None    # small: always built
    csr = _csr_cpu(sps.csr_matrix((1, 4)), np.float64)
    csr.n = 1 << 28
    assert SlabEnt.from_csr(csr) is None


# ------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _ent_vs_oracle(n, m, k, density, dtype, seed, d_zero_every=0, want_colsum=False, poison=False):
    from oracle import oracle as orc
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import DenseDev

    rng = np.random.default_rng(seed)
    S = sps.random(n, m, density=density, format="csr", random_state=rng, dtype=np.float64)
    S.data = rng.standard_normal(S.data.shape[0])
    S = S.astype(dtype)
    S.sort_indices()
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    if d_zero_every:
        d[::d_zero_every] = 0
        if poison:                      # a row with d == 0 may hold anything: inf * 0 must not leak
            B[::d_zero_every] = np.inf
    csr = CsrDev(torch.from_numpy(S.data.copy()).cuda(), torch.from_numpy(S.indices.astype(np.int32)).cuda(),
                 torch.from_numpy(S.indptr.astype(np.int64)).cuda(), n, m)
    tw = SlabEnt.from_csr(csr)
    Bd = DenseDev(torch.from_numpy(B).cuda(), n, k, 0)
    res = xs.csr_dense_sandwich_ent(tw, Bd, torch.from_numpy(d).cuda(), want_colsum=want_colsum)
    Bref = B.astype(np.float64)
    if poison:
        Bref[::d_zero_every] = 0.0
    ref = orc.csr_dense_sandwich(S.astype(np.float64).tocsr(), Bref, d.astype(np.float64), None, None, None)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    out = res[0] if want_colsum else res
    scale = max(np.abs(ref).max(), 1e-300)
    assert np.abs(out.cpu().numpy().astype(np.float64) - ref).max() / scale < tol
    if want_colsum:
        cref = S.astype(np.float64).T @ d.astype(np.float64)
        assert np.abs(res[1].cpu().numpy() - cref).max() / max(np.abs(cref).max(), 1e-300) < tol


@gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,k,density", [
    (1000, 40, 128, 0.3),        # many batches per block
    (5003, 100, 136, 0.05),      # ragged last slab, ragged second dense part
    (70, 16, 128, 0.9),          # one group, almost dense
    (20011, 512, 256, 0.02),     # two workgroups of column groups, two dense parts
    (64 * 300 + 1, 300, 128, 0.004),   # most blocks empty: slabs without a batch
    (3, 1, 128, 1.0),
])
def test_ent_kernel_vs_oracle(n, m, k, density, dtype):
    _ent_vs_oracle(n, m, k, density, dtype, seed=n + m)


@gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ent_kernel_zero_weights_and_colsum(dtype):
    _ent_vs_oracle(9000, 200, 128, 0.05, dtype, seed=5, d_zero_every=3, want_colsum=True)
    _ent_vs_oracle(9000, 200, 128, 0.05, dtype, seed=6, d_zero_every=5, want_colsum=False, poison=True)


def _structured(kind, n, m, rng):
    """Sparsity patterns a uniform random matrix never shows."""
    if kind == "one_full_column":              # every row has an entry in one column: blocks of 64 slots, every slab
        S = sps.lil_matrix((n, m))
        S[:, m // 2] = rng.standard_normal((n, 1))
        S[rng.integers(0, n, 50), rng.integers(0, m, 50)] = 1.5
    elif kind == "last_group_last_slab":       # entries only in the last 16-column group of the last (partial) slab
        S = sps.lil_matrix((n, m))
        r0 = (n - 1) // 64 * 64
        for r in range(r0, n):
            S[r, m - 1 - (r % min(16, m))] = float(r + 1)
    elif kind == "banded":                     # column ~ row: consecutive slabs hit consecutive groups
        rows = np.arange(n)
        S = sps.csr_matrix((rng.standard_normal(3 * n), (np.repeat(rows, 3),
                            (np.repeat(rows * m // n, 3) + np.tile([0, 1, 2], n)) % m)), shape=(n, m)).tolil()
    elif kind == "dense_rows":                 # a few completely filled rows, the rest empty
        S = sps.lil_matrix((n, m))
        for r in rng.choice(n, 5, replace=False):
            S[r, :] = rng.standard_normal((1, m))
    elif kind == "far_apart_slabs":            # entries in a few slabs hundreds of slabs apart (round 6: the 16-bit meta
        S = sps.lil_matrix((n, m))             # word carries 6 bits of the slab; empty stretches get continuity batches)
        for s0 in sorted(set([0, (n // 64) // 3, max(0, (n - 1) // 64 - 70), (n - 1) // 64])):
            for r in range(s0 * 64, min(s0 * 64 + 64, n), 3):
                S[r, (7 * r) % m] = float(r % 11 + 1)
                S[r, (7 * r + m // 2) % m] = -0.5
    else:                                      # "first_slab_only"
        S = sps.lil_matrix((n, m))
        S[:min(64, n), :] = rng.standard_normal((min(64, n), m))
    S = S.tocsr()
    S.sum_duplicates()
    S.sort_indices()
    return S


@gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["one_full_column", "last_group_last_slab", "banded", "dense_rows", "first_slab_only",
                                  "far_apart_slabs"])
@pytest.mark.parametrize("n,m", [(5003, 100), (64 * 40, 512), (130, 17), (64 * 700 + 5, 48)])
def test_ent_kernel_structured_patterns(kind, n, m, dtype):
    """Blocks of exactly / more than 64 slots in every slab, groups and slabs without any entry, entries only in the
    ragged tail: the cases the superbatch fold, the quad cut and the slab walk have branches for."""
    from oracle import oracle as orc
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import DenseDev

    rng = np.random.default_rng(n + m + len(kind))
    S = _structured(kind, n, m, rng).astype(dtype)
    k = 128
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    csr = CsrDev(torch.from_numpy(S.data.copy()).cuda(), torch.from_numpy(S.indices.astype(np.int32)).cuda(),
                 torch.from_numpy(S.indptr.astype(np.int64)).cuda(), n, m)
    tw = SlabEnt.from_csr(csr, max_pad=1e9)
    assert tw is not None
    out, cs = xs.csr_dense_sandwich_ent(tw, DenseDev(torch.from_numpy(B).cuda(), n, k, 0), torch.from_numpy(d).cuda(),
                                        want_colsum=True)
    ref = orc.csr_dense_sandwich(S.astype(np.float64).tocsr(), B.astype(np.float64), d.astype(np.float64),
                                 None, None, None)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert np.abs(out.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300) < tol
    cref = S.astype(np.float64).T @ d.astype(np.float64)
    assert np.abs(cs.cpu().numpy() - cref).max() / max(np.abs(cref).max(), 1e-300) < tol


@gpu
def test_ent_kernel_explicit_zero_values_and_empty_matrix():
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import DenseDev

    n, m, k = 500, 32, 128
    rng = np.random.default_rng(1)
    S = sps.random(n, m, density=0.2, format="csr", random_state=rng, dtype=np.float64)
    S.data[::4] = 0.0                    # stored zeros stay harmless
    S.sort_indices()
    B = rng.standard_normal((n, k))
    d = rng.random(n)
    for mat in (S, sps.csr_matrix((n, m))):
        csr = CsrDev(torch.from_numpy(mat.data.copy()).cuda(), torch.from_numpy(mat.indices.astype(np.int32)).cuda(),
                     torch.from_numpy(mat.indptr.astype(np.int64)).cuda(), n, m)
        tw = SlabEnt.from_csr(csr)
        out = xs.csr_dense_sandwich_ent(tw, DenseDev(torch.from_numpy(B).cuda(), n, k, 0), torch.from_numpy(d).cuda())
        ref = mat.T @ (d[:, None] * B)
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0)


@gpu
def test_sparse_matrix_takes_the_entry_kernel():
    """SparseMatrix x DenseMatrix of more than 64 C-ordered columns runs on the entry twin."""
    import tabmat_amd as tm
    from tabmat_amd import _lib

    rng = np.random.default_rng(2)
    n = 4000
    S = sps.random(n, 100, density=0.05, format="csc", random_state=rng)
    X = rng.standard_normal((n, 128))
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(X)
    d = rng.random(n)
    seen = []
    orig = _lib.call

    def spy(name, *a):
        seen.append(name)
        return orig(name, *a)

    import tabmat_amd.ext.sparse as xs_mod
    xs_mod.call = spy
    try:
        out = sm._cross_sandwich(dm, d, None, None, None)
    finally:
        xs_mod.call = orig
    assert any(s.startswith("tm_csr_dense_sandwich_ent_") for s in seen), seen
    ref = S.T @ (d[:, None] * X)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-12


@gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_deterministic_sparse_self_sandwich(dtype, monkeypatch):
    """TABMAT_AMD_DETERMINISTIC: the sparse self sandwich in a fixed summation order (column chunks through the
    entry-list kernel) -- bit-identical from run to run, exactly symmetric, equal to the oracle within the
    float tolerance; the reference's kernel is deterministic by construction (ext/sparse.pyx:55-74)."""
    import tabmat_amd as tm
    import tabmat_amd.categorical_matrix as cmod
    from oracle import oracle as orc

    rng = np.random.default_rng(12)
    n, m = 30_011, 300
    S = sps.random(n, m, density=0.05, format="csc", random_state=rng, dtype=np.float64).astype(dtype)
    d = rng.random(n).astype(dtype)
    sm = tm.SparseMatrix(S)
    monkeypatch.setattr(cmod, "DETERMINISTIC", True)
    a = sm.sandwich(d)
    b = sm.sandwich(d)
    assert np.array_equal(a, b)
    assert np.array_equal(a, a.T)
    want = orc.sparse_sandwich(sps.csc_matrix(S).astype(np.float64), sps.csr_matrix(S).astype(np.float64),
                               d.astype(np.float64), None, None)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    assert np.abs(a - want).max() / np.abs(want).max() < tol
    rows = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
    cols = np.sort(rng.choice(m, 50, replace=False)).astype(np.int32)
    Sr = S.tocsr()[rows][:, cols].astype(np.float64)
    want_rc = (Sr.T.multiply(d[rows].astype(np.float64))).dot(Sr).toarray()
    got_rc = sm.sandwich(d, rows, cols)
    assert np.abs(got_rc - want_rc).max() / np.abs(want_rc).max() < tol
    assert np.array_equal(got_rc, sm.sandwich(d, rows, cols))
    # a column id twice inside one 128-column window, unsorted: the product of X[:, cols], as the
    # LDS-atomic path gives it -- the repeats are expanded from the product over the distinct columns
    cols_dup = np.concatenate([cols[:20][::-1], cols[5:12], cols[30:]]).astype(np.int32)
    Sd = S.tocsr()[rows][:, cols_dup].astype(np.float64)
    want_dup = (Sd.T.multiply(d[rows].astype(np.float64))).dot(Sd).toarray()
    got_dup = sm.sandwich(d, rows, cols_dup)
    assert np.abs(got_dup - want_dup).max() / np.abs(want_dup).max() < tol
    monkeypatch.setattr(cmod, "DETERMINISTIC", False)
    assert np.abs(sm.sandwich(d, rows, cols_dup) - want_dup).max() / np.abs(want_dup).max() < tol
    c = sm.sandwich(d)
    assert np.abs(a - c).max() / np.abs(want).max() < tol


@gpu
def test_deterministic_mode_builds_its_own_twin_for_a_very_sparse_block(monkeypatch):
    """a block too sparse for the product path's entry twin (padded stream beyond ELL_MAX_PAD x the
    nonzeros) used to fall through silently to the LDS-atomic kernels under TABMAT_AMD_DETERMINISTIC=1; the mode
    now builds a twin without that limit, and says so (RuntimeWarning) when no twin can exist at all."""
    import warnings

    import tabmat_amd as tm
    import tabmat_amd.categorical_matrix as cmod
    import tabmat_amd.sparse_matrix as smod

    rng = np.random.default_rng(13)
    n, m = 2_400_000, 512
    S = sps.random(n, m, density=0.0005, format="csc", random_state=rng)
    d = rng.random(n)
    sm = tm.SparseMatrix(S)
    # ~600k nonzeros in ~500k (slab, group) blocks of >= 16 slots each: > 8 x nnz slots and > the 4M-slot floor
    assert sm._ent() is None                       # refused for the products (SlabEnt.from_csr)
    monkeypatch.setattr(cmod, "DETERMINISTIC", True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = sm.sandwich(d)
        b = sm.sandwich(d)
    assert sm._ent_det() is not None
    assert np.array_equal(a, b) and np.array_equal(a, a.T)
    want = (S.T.multiply(d)).dot(S).toarray()
    assert np.abs(a - want).max() / np.abs(want).max() < 1e-12
    cols = np.sort(rng.choice(m, 40, replace=False)).astype(np.int32)
    assert np.array_equal(sm.sandwich(d, None, cols), a[np.ix_(cols, cols)])
    # no twin at all -> a warning, not silence
    monkeypatch.setattr(smod.SparseMatrix, "_ent_det", lambda self: None)
    with pytest.warns(RuntimeWarning, match="NOT bit-reproducible"):
        sm.sandwich(d)


@pytest.fixture
def _catsparse_kernel(request):
    """Both kernels behind tm_multi_cat_sparse_sandwich_ent*: "staged" (round 6: d and the code words of a slab's rows
    parked in per-wave LDS; the default wherever the LDS holds tile + staging) and "gather" (round 4 / 5: per-slot
    gathers; the fallback)."""
    from tabmat_amd import _lib

    _lib.call("tm_tune_set", b"catsparse_staged", 1 if request.param == "staged" else 0)
    _lib.call("tm_tune_set", b"catsparse_staged_fill", 0)          # (whatever the blocks hold: the kernel is forced)
    yield request.param
    _lib.call("tm_tune_set", b"catsparse_staged", -2**63)
    _lib.call("tm_tune_set", b"catsparse_staged_fill", -2**63)


@gpu
@pytest.mark.parametrize("_catsparse_kernel", ["staged", "gather"], indirect=True)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,levels", [(9000, 100, (7, 5)), (20_011, 512, (256, 96, 32)), (70, 16, (3,)),
                                        (30_000, 40, (40, 30, 20, 10, 5, 4, 3, 2)), (64 * 300 + 1, 48, (11, 6, 2))])
def test_cat_sparse_cross_terms_on_the_entry_twin(n, m, levels, dtype, _catsparse_kernel):
    """tm_multi_cat_sparse_sandwich_ent_*: all categorical x sparse blocks of a SplitMatrix from one pass over the
    entry twin, against the oracle's sandwich_cat_sparse (the reference: scipy.sparse product,
    categorical_matrix.py:825-838); drop_first, missing codes, zero weights; a last slab of ONE row; dense columns
    whose blocks hold more than 64 slots (the staged kernel's second step)."""
    from oracle import oracle as orc
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=0.06, format="csr", random_state=rng, dtype=np.float64)
    if m >= 40:              # five nearly full columns: their groups' blocks hold well over 64 slots per slab
        extra = sps.random(n, m, density=1.0, format="csr", random_state=rng, dtype=np.float64).multiply(
            sps.csr_matrix(np.isin(np.arange(m), [1, 2, 17, 18, 33]).astype(np.float64))).tocsr()
        keep = sps.csr_matrix((rng.random(n) < 0.85).astype(np.float64)[:, None])
        S = (S + extra.multiply(keep)).tocsr()
    S.data = rng.standard_normal(S.data.shape[0])
    S = S.astype(dtype)
    S.sort_indices()
    d = rng.random(n).astype(dtype)
    d[::6] = 0
    csr = CsrDev(torch.from_numpy(S.data.copy()).cuda(), torch.from_numpy(S.indices.astype(np.int32)).cuda(),
                 torch.from_numpy(S.indptr.astype(np.int64)).cuda(), n, m)
    tw = SlabEnt.from_csr(csr)
    cats, refs = [], []
    for k, lv in enumerate(levels):
        codes = rng.integers(0, lv, n).astype(np.int32)
        if k % 2:
            codes[rng.random(n) < 0.05] = -1
        drop = bool(k % 3 == 1)
        cats.append((torch.from_numpy(codes).cuda(), lv - int(drop), drop))
        refs.append(orc.sandwich_cat_sparse(codes, lv, d.astype(np.float64), S.astype(np.float64).tocsr(), None, None,
                                            None)[int(drop):])
    got = xsplit.multi_cat_sparse_sandwich_ent(cats, torch.from_numpy(d).cuda(), tw).cpu().numpy()
    want = np.vstack(refs)
    assert got.shape == want.shape
    tol = 1e-10 if dtype == np.float64 else 3e-5
    assert np.abs(got - want).max() / max(np.abs(want).max(), 1e-300) < tol
    # round 5: the codes of up to 3 categoricals packed into one word per row (tm_multi_cat_pack_codes / _entp_)
    pk = xsplit.pack_codes(cats)
    assert (pk is not None) == (len(levels) <= 3)
    if pk is not None:
        pkh = pk.cpu().numpy().view(np.uint32)
        off = 0
        for k, (ct, ncol, drop) in enumerate(cats):
            col = ct.cpu().numpy().astype(np.int64) - int(drop)
            want_f = np.where(col >= 0, off + col, 1023)
            assert np.array_equal((pkh >> (10 * k)) & 1023, want_f)
            off += ncol
        got_p = xsplit.multi_cat_sparse_sandwich_ent(cats, torch.from_numpy(d).cuda(), tw, pk).cpu().numpy()
        assert np.abs(got_p - want).max() / max(np.abs(want).max(), 1e-300) < tol


@gpu
def test_split_matrix_needs_no_slab_twin_beside_the_entry_twin():
    """A SplitMatrix with a wide C-ordered dense block: sparse x dense AND categorical x sparse run on the entry
    twin; the slab-form twin of the sparse block is never built (3.4 GB at BASELINE configs[3])."""
    import tabmat_amd as tm

    rng = np.random.default_rng(3)
    n = 20_000
    X = rng.standard_normal((n, 128))
    S = sps.random(n, 200, density=0.05, format="csc", random_state=rng)
    c1, c2 = rng.integers(0, 20, n), rng.integers(0, 7, n)
    mat = tm.SplitMatrix([tm.DenseMatrix(X), tm.SparseMatrix(S), tm.CategoricalMatrix(c1), tm.CategoricalMatrix(c2)])
    d = rng.random(n)
    E = np.hstack([X, S.toarray(), np.eye(20)[c1], np.eye(7)[c2]])
    for m2 in (mat, tm.SplitMatrix([tm.DenseMatrix(X), tm.SparseMatrix(S), tm.CategoricalMatrix(c1),
                                    tm.CategoricalMatrix(c2)]).to_device()):
        got = m2.sandwich(d)
        want = E.T @ (d[:, None] * E)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-10
        sp = m2.matrices[1]
        assert getattr(sp, "_entblk", None) and getattr(sp, "_slabblk", None) is None


@gpu
@pytest.mark.parametrize("_catsparse_kernel", ["staged", "gather"], indirect=True)
def test_cat_sparse_on_slabs_far_apart(_catsparse_kernel):
    """Round 6 (16-bit meta word): blocks hundreds of slabs apart, last slab ragged -- the gather kernel rebuilds every
    slot's slab from its 6-bit tag and a running slab, the staged kernel walks the slabs themselves; both against the
    oracle."""
    from oracle import oracle as orc
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(3)
    n, m = 64 * 700 + 5, 48
    S = _structured("far_apart_slabs", n, m, rng)
    d = rng.random(n)
    csr = CsrDev(torch.from_numpy(S.data.copy()).cuda(), torch.from_numpy(S.indices.astype(np.int32)).cuda(),
                 torch.from_numpy(S.indptr.astype(np.int64)).cuda(), n, m)
    tw = SlabEnt.from_csr(csr, max_pad=1e9)
    cats, refs = [], []
    for lv in (9, 4):
        codes = rng.integers(0, lv, n).astype(np.int32)
        cats.append((torch.from_numpy(codes).cuda(), lv, False))
        refs.append(orc.sandwich_cat_sparse(codes, lv, d, S.tocsr(), None, None, None))
    for pk in (None, xsplit.pack_codes(cats)):
        got = xsplit.multi_cat_sparse_sandwich_ent(cats, torch.from_numpy(d).cuda(), tw, pk).cpu().numpy()
        want = np.vstack(refs)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
