"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol that
include/tabmat_hip.h declares (no compute calls), the host-side mirror reproduces the reference's
argument handling and error conventions (util.py:27-67), and the product path fails LOUDLY
without a GPU instead of falling back to the CPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs
import tabmat_amd as tm
from tabmat_amd import _lib
from tabmat_amd.ext import split as xsplit

HAS_GPU = torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    protos = _lib.prototypes()
    assert len(protos) >= 50
    lib = _lib.lib()          # raises AttributeError on a missing symbol
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    for name in protos:
        assert f" T {name}" in exported, name
        assert getattr(lib, name).argtypes is not None
    assert lib.tm_version() >= 100
    assert isinstance(_lib.last_error(), str)


def test_header_prototypes_are_plain_c():
    src = open(_lib.HEADER).read()
    import re
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)   # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code
    assert 'extern "C"' in src
    for name, args in _lib.prototypes().items():
        assert all(a in (C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float) for a in args), name


def test_no_cpu_fallback():
    """Without a GPU every product raises; construction and validation still work."""
    if HAS_GPU:
        pytest.skip("GPU present")
    X = tm.DenseMatrix(np.ones((4, 2)))
    with pytest.raises(_lib.TabmatHipError, match="no CPU fallback"):
        X.sandwich(np.ones(4))
    with pytest.raises(_lib.TabmatHipError):
        tm.CategoricalMatrix([0, 1, 1]).transpose_matvec(np.ones(3))
    import inspect
    import tabmat_amd

    for mod in ("dense_matrix", "sparse_matrix", "categorical_matrix", "split_matrix", "_lib"):
        text = inspect.getsource(getattr(tabmat_amd, mod))
        assert "oracle" not in text, f"product module {mod} must not touch the oracle"


def _mats():
    specs, idx = cs.complex_split_specs()
    from _gpu_util import to_tm_block, to_tm_split

    return [to_tm_block(s) for _, s in cs.unscaled_specs()] + [to_tm_split(specs, idx)]


@pytest.mark.parametrize("k", range(8))
def test_error_conventions_before_any_device_work(k):
    """tests/test_matrices.py:110-126,174-216: exceptions by type and message, raised by the
    host-side checks (so they are testable without a GPU)."""
    mat = _mats()[k]
    n, m = mat.shape
    for bad in (n - 1, n + 1):
        with pytest.raises(ValueError, match="not aligned"):
            mat.sandwich(np.ones(bad))
        with pytest.raises(ValueError):
            mat.transpose_matvec(np.ones(bad))
    for bad in (m - 1, m + 1):
        with pytest.raises(ValueError):
            mat.matvec(np.ones(bad))
    with pytest.raises(TypeError, match="same dtype"):
        mat.astype(np.float64).sandwich(np.ones(n, dtype=np.float32))
    with pytest.raises(ValueError, match="first dimension of 'out' must be"):
        mat.matvec(np.zeros(m), None, np.zeros(n + 1))
    with pytest.raises(ValueError, match="dimension of 'out' must be"):
        mat.transpose_matvec(np.zeros(n), None, None, np.zeros(m + 1))


def test_categorical_2d_not_implemented():
    c = tm.CategoricalMatrix([0, 1, 1])
    with pytest.raises(NotImplementedError, match="only implemented for 1d"):
        c.matvec(np.ones((2, 2)))
    with pytest.raises(NotImplementedError, match="only implemented for 1d"):
        c.transpose_matvec(np.ones((3, 2)))


def test_split_matrix_construction_rules():
    """split_matrix.py:171-267: merging of dense / sparse blocks, index validation."""
    specs = [s for _, s in cs.unscaled_specs()]
    from _gpu_util import to_tm_block

    X = tm.SplitMatrix([to_tm_block(s) for s in specs])
    kinds = [type(m).__name__ for m in X.matrices]
    assert kinds == ["DenseMatrix", "SparseMatrix", "CategoricalMatrix", "CategoricalMatrix"]
    assert X.shape == (3, 14)
    assert [i.dtype for i in X.indices] == [np.int64] * 4
    ref_specs, ref_idx = cs.combine_specs(specs)
    for a, b in zip(X.indices, ref_idx):
        assert np.array_equal(a, b)
    np.testing.assert_array_equal(X.toarray(), np.hstack([cs.spec_toarray(s) for s in specs]))
    with pytest.raises(ValueError, match="sorted"):
        tm.SplitMatrix([tm.DenseMatrix(np.random.random((10, 3)))], [[1, 0, 2]])
    with pytest.raises(ValueError, match="same first dimension"):
        tm.SplitMatrix([tm.DenseMatrix(np.ones((3, 1))), tm.DenseMatrix(np.ones((4, 1)))])
    # nested SplitMatrix is flattened (tests/test_split_matrix.py:136-141)
    np.testing.assert_array_equal(tm.SplitMatrix([X, X]).toarray(),
                                  np.hstack([X.toarray(), X.toarray()]))


@pytest.mark.parametrize("seed", range(5))
def test_split_col_subsets_matches_oracle(seed):
    """ext/split.pyx:157-209, host-side index bookkeeping."""
    from oracle import oracle as orc

    rng = np.random.default_rng(seed)
    p = 40
    perm = rng.permutation(p)
    cuts = np.sort(rng.choice(np.arange(1, p), size=3, replace=False))
    indices = [np.sort(part) for part in np.split(perm, cuts)]

    class Fake:
        pass

    f = Fake()
    f.indices = [np.asarray(i, dtype=np.int64) for i in indices]
    cols = np.sort(rng.choice(p, size=17, replace=False)).astype(np.int32)
    a, b, n = xsplit.split_col_subsets(f, cols)
    ra, rb, rn = orc.split_col_subsets(f.indices, cols)
    assert n == rn == 17
    for x, y in zip(a + b, ra + rb):
        assert np.array_equal(x, y)
    for i in range(len(indices)):
        assert np.array_equal(f.indices[i][b[i]], cols[a[i]])


def test_categorical_construction():
    """categorical_matrix.py:351-430: codes, missing handling, drop_first shape."""
    c = tm.CategoricalMatrix(["b", "a", "b", None], cat_missing_method="zero")
    assert c.indices.dtype == np.int32 and list(c.indices) == [1, 0, 1, -1]
    assert c.shape == (4, 2) and c._has_missings
    with pytest.raises(ValueError, match="missing values"):
        tm.CategoricalMatrix(["b", None])
    conv = tm.CategoricalMatrix(["b", "a", None], cat_missing_method="convert")
    assert conv.shape == (3, 3) and list(conv.indices) == [1, 0, 2]
    d = tm.CategoricalMatrix([0, 1, 2], drop_first=True)
    assert d.shape == (3, 2)
    np.testing.assert_array_equal(d.toarray(), [[0, 0], [1, 0], [0, 1]])
    with pytest.raises(ValueError, match="exceed"):
        tm.CategoricalMatrix([0, 5], categories=np.arange(3))


def test_sparse_matrix_construction():
    """sparse_matrix.py:35-79: CSC, sorted indices, common index dtype."""
    S = sps.random(30, 7, density=0.3, format="coo", random_state=1)
    m = tm.SparseMatrix(S)
    assert m.array_csc.has_sorted_indices and m.shape == (30, 7)
    assert m.indices.dtype == m.indptr.dtype
    np.testing.assert_allclose(m.toarray(), S.toarray())
    assert m.array_csr.shape == (30, 7)


def test_chunk_major_twin_round_trip():
    """Host-side ingest of the chunk-major twin K2 streams (tabmat_amd/ext/_types.py
    CsrDev.chunk_major): every 128-column chunk is a CSR matrix of its own, rows adjacent, column
    order kept (pure host logic on CPU tensors; the kernel consuming it is tested under -m gpu)."""
    import scipy.sparse as sps
    import torch

    from tabmat_amd._lib import lib
    from tabmat_amd.ext._types import CsrDev

    n, m = 300, 300
    A = sps.random(n, m, density=0.07, format="csr", random_state=5, dtype=np.float64)
    A.sort_indices()
    csr = CsrDev(torch.from_numpy(A.data), torch.from_numpy(A.indices.astype(np.int32)),
                 torch.from_numpy(A.indptr.astype(np.int64)), n, m)
    data, ind, cptr = (t.numpy() for t in csr.chunk_major())
    ch = lib().tm_sparse_chunk_cols()
    nch = (m + ch - 1) // ch
    assert cptr.shape == (nch, n + 1) and cptr.dtype == np.int32
    assert cptr[0, 0] == 0 and cptr[-1, -1] == A.nnz
    B = np.zeros((n, m))
    for c in range(nch):
        assert c == 0 or cptr[c, 0] == cptr[c - 1, n]          # chunks follow each other
        for k in range(n):
            lo, hi = cptr[c, k], cptr[c, k + 1]
            assert ind.dtype == np.uint8            # round 6: the column INSIDE the chunk, one byte per entry
            cols = c * ch + ind[lo:hi].astype(np.int64)
            assert np.all(cols < m) and np.all(np.diff(cols) > 0)
            B[k, cols] += data[lo:hi]
    np.testing.assert_array_equal(B, A.toarray())
    # the int32 block columns are rebuilt on request (A/B switch, record twins), not kept
    c32 = csr.chunk_cols32().numpy()
    assert c32.dtype == np.int32
    for c in range(nch):
        seg = slice(cptr[c, 0], cptr[c, n])
        np.testing.assert_array_equal(c32[seg], c * ch + ind[seg].astype(np.int32))
    # 16-bit CSR columns after compaction; `.indices` widens them again (shared scratch)
    assert csr.compact_indices() and csr._ind32 is None
    np.testing.assert_array_equal(csr.indices.numpy(), A.indices.astype(np.int32))
    wide = CsrDev(torch.from_numpy(A.data), torch.from_numpy((A.indices.astype(np.int64) * 200).astype(np.int32)),
                  torch.from_numpy(A.indptr.astype(np.int64)), n, m * 200)
    wide.compact_indices()
    np.testing.assert_array_equal(wide.indices.numpy(), (A.indices.astype(np.int64) * 200).astype(np.int32))
    np.testing.assert_array_equal(csr.indices.numpy(), A.indices.astype(np.int32))     # (scratch changed hands)


def test_slab_stream_round_trip():
    """Host-side ingest of the slab-blocked stream of the gather / cat x sparse kernels
    (tabmat_amd/ext/_types.py SlabCsc): runs per (slab, column), group pointers, row offsets and
    the in-group column of every entry reproduce the matrix (pure host logic on CPU tensors)."""
    import scipy.sparse as sps
    import torch

    from tabmat_amd._lib import lib
    from tabmat_amd.ext._types import CsrDev, SlabCsc

    n, m = 700, 70
    A = sps.random(n, m, density=0.08, format="csr", random_state=3, dtype=np.float64)
    A.sort_indices()
    csr = CsrDev(torch.from_numpy(A.data), torch.from_numpy(A.indices.astype(np.int32)),
                 torch.from_numpy(A.indptr.astype(np.int64)), n, m)
    S = SlabCsc.from_csr(csr)
    R, C = lib().tm_slab_rows(), lib().tm_slab_group_cols()
    G = (m + C - 1) // C
    mpad = G * C
    ns = (n + R - 1) // R
    vals, koff, ecol = S.vals.numpy(), S.koff.numpy(), S.ecol.numpy()
    cnt = S.cnt.numpy().astype(np.uint16).reshape(ns, mpad)
    gptr = S.gptr.numpy()
    assert len(vals) == A.nnz and gptr[-1] == A.nnz
    B = np.zeros((n, m))
    pos = 0
    for s in range(ns):
        for c in range(mpad):
            if c % C == 0:
                assert gptr[s * G + c // C] == pos
            last = -1
            for e in range(pos, pos + int(cnt[s, c])):
                assert ecol[e] == c % C
                r = koff[e] // (64 * 8)
                assert r > last                      # rows ascending inside a run
                last = r
                B[s * R + r, c] += vals[e]
            pos += int(cnt[s, c])
    np.testing.assert_array_equal(B, A.toarray())


@pytest.mark.parametrize("wide", [False, True])
def test_slab_ell_round_trip(wide):
    """Host-side ingest of the interleaved-ELL twin of the static gather kernel
    (tabmat_amd/ext/_types.py SlabEll): iterations, padding, group pointers, row offsets and the
    column permutation reproduce the matrix (pure host logic on CPU tensors); wide = the geometry
    of tm_csr_dense_sandwich_ellw_* (64-row slabs, 16 columns x 4 slots, 128-column LDS rows)."""
    import scipy.sparse as sps
    import torch

    from tabmat_amd._lib import lib
    from tabmat_amd.ext._types import CsrDev, SlabEll

    n, m = 700, 70
    A = sps.random(n, m, density=0.08, format="csr", random_state=3, dtype=np.float64)
    A.sort_indices()
    csr = CsrDev(torch.from_numpy(A.data), torch.from_numpy(A.indices.astype(np.int32)),
                 torch.from_numpy(A.indptr.astype(np.int64)), n, m)
    E = SlabEll.from_csr(csr, wide=wide)
    assert E.wide == wide
    if wide:
        R, C, W = lib().tm_ellw_rows(), lib().tm_ellw_group_cols(), 128
        assert (R, C) == (64, 16)
    else:
        R, C, W = lib().tm_slab_rows(), lib().tm_slab_group_cols(), 64
    U = 64 // C
    G = (m + C - 1) // C
    assert E.mk == G * C
    ns = (n + R - 1) // R
    vals, koff, gptr, inv = E.vals.numpy(), E.koff.numpy(), E.gptr.numpy(), E.inv.numpy()
    assert sorted(inv.tolist()) == list(range(m))
    kcol_to_col = {int(inv[c]): c for c in range(m)}
    B = np.zeros((n, m))
    n_real = 0
    for s in range(ns):
        for g in range(G):
            b0, b1 = gptr[s * G + g], gptr[s * G + g + 1]
            assert (b1 - b0) % 64 == 0
            longest, last = 0, {}
            for e in range(b0, b1):
                it, rem = divmod(e - b0, 64)
                c, u = divmod(rem, U)
                if koff[e] == -1:
                    assert vals[e] == 0
                    continue
                kc = g * C + c
                assert koff[e] % (W * 8) == 0
                r = koff[e] // (W * 8)
                assert 0 <= r < R
                assert r > last.get(kc, -1)          # rows ascending along (it, u)
                last[kc] = r
                B[s * R + r, kcol_to_col[kc]] += vals[e]
                longest = max(longest, it * U + u + 1)
                n_real += 1
            assert (b1 - b0) // 64 == (longest + U - 1) // U   # padded to the longest run only
    assert gptr[-1] == len(vals) and n_real == A.nnz
    np.testing.assert_array_equal(B, A.toarray())


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("diag_only", [False, True])
def test_cat_pairs_plan_layout(seed, diag_only):
    """The bundling of the categorical x categorical tables (tabmat_amd/ext/split.py::CatPairsPlan):
    every table that fits a tile appears exactly once, tiles stay within the LDS budget and the slot
    count, tables of one bundle do not overlap, the workgroup map is consistent."""
    from tabmat_amd._lib import lib

    cap = int(lib().tm_multi_cat_pairs_max_bins())
    slots = int(lib().tm_multi_cat_pairs_max_slots())
    rng = np.random.default_rng(seed)
    k = int(rng.choice([1, 2, 3, 7, 20, 45, 100]))
    pool = [1, 2, 3, 5, 12, 50, 100, 127, 128, 129, 300, 2000, cap, cap + 1, 40_000]
    L = [int(rng.choice(pool)) for _ in range(k)]
    plan = xsplit.CatPairsPlan([(10 + i, L[i]) for i in range(k)], None, diag_only=diag_only)
    B, desc = plan.h_bundles, plan.h_desc
    assert plan.n_bundles == 0 or B.shape[0] == plan.n_bundles
    assert plan.slots <= slots and plan.bins <= cap
    seen = {}
    for (bi, bj, off, li, lj, stride), q in zip(plan.pairs, desc):
        a, b = bi - 10, bj - 10
        assert a <= b and (li, lj) == (L[a], L[b]) and (a, b) not in seen
        seen[(a, b)] = (off, stride)
        assert q[0] == off and q[1] == li and q[2] == lj and q[3] == stride and q[6] == int(a == b)
    if diag_only:
        assert set(seen) == {(a, a) for a in range(k) if L[a] <= cap}
    else:
        for a in range(k):
            assert ((a, a) in seen) == (L[a] <= cap)
            for b in range(a + 1, k):
                assert ((a, b) in seen) == (L[a] * L[b] <= cap), (L[a], L[b])
    # per bundle: slot counts, tile size, disjoint tables
    wg = plan.h_wg_map
    for y in range(plan.n_bundles):
        na, nb, mode, width, wg0, parts, bins = (int(x) for x in B[y][:7])
        assert 1 <= na <= slots and 1 <= nb <= slots and bins <= plan.bins <= cap and parts >= 1
        assert (wg[wg0:wg0 + parts] == y).all() and (wg0 == 0 or wg[wg0 - 1] == y - 1)
        A = [(int(B[y][8 + 2 * s]), int(B[y][9 + 2 * s])) for s in range(na)]
        Bs = [(int(B[y][8 + 2 * slots + 2 * s]), int(B[y][9 + 2 * slots + 2 * s])) for s in range(nb)]
        used = np.zeros(bins, dtype=np.int32)
        for s, (a, ya) in enumerate(A):
            for t, (b, xb) in enumerate(Bs):
                if (mode == 1 and t < s) or (mode == 2 and t != s):
                    continue
                rows_ = ya + np.arange(L[a])[:, None] * width
                cols_ = (xb + np.arange(L[b])[None, :]) if mode != 2 else np.zeros((1, 1), dtype=int)
                if mode == 2:
                    idx = (ya + np.arange(L[a]) * width)
                elif mode == 1 and s == t:
                    idx = (rows_ + cols_).ravel()
                else:
                    idx = (rows_ + cols_).ravel()
                assert idx.min() >= 0 and idx.max() < bins
                used[idx] += 1
                # the scatter descriptor addresses the same bins
                off, stride = seen[(min(a, b), max(a, b))] if a != b else seen[(a, a)]
                if a == b:
                    d_idx = off - y * plan.bins + np.arange(L[a]) * stride
                    assert np.isin(d_idx, idx).all()
                else:
                    assert off - y * plan.bins == ya + xb and stride == width
        assert used.max() <= 1
    assert len(wg) == plan.n_wg or plan.n_wg == 0


@pytest.mark.parametrize("seed", range(8))
def test_split_col_subsets_table_form_equals_merge_loop(seed):
    """The table-lookup form of split_col_subsets (strictly increasing selections) and the merge
    loop it replaces (ext/split.pyx:157-209) give the same lists; unsorted / repeated selections
    fall back to the loop; the identity selection collapses to None."""
    from tabmat_amd.util import collapse_identity

    rng = np.random.default_rng(100 + seed)
    p = int(rng.choice([1, 5, 64, 300]))
    perm = rng.permutation(p)
    k = int(rng.integers(1, min(p, 6) + 1))
    cuts = np.sort(rng.choice(np.arange(1, p), size=k - 1, replace=False)) if p > 1 and k > 1 else []
    indices = [np.sort(part).astype(np.int64) for part in np.split(perm, cuts)]

    class Fake:
        pass

    f = Fake()
    f.indices = indices
    g = Fake()
    g.indices = indices
    g.__dict__["_col_maps"] = None          # forces the merge loop
    for size in {1, max(1, p // 3), p}:
        cols = np.sort(rng.choice(p, size=size, replace=False)).astype(np.int32)
        a = xsplit.split_col_subsets(f, cols)
        b = xsplit.split_col_subsets(g, cols)
        assert a[2] == b[2]
        for x, y in zip(a[0] + a[1], b[0] + b[1]):
            assert np.array_equal(x, y)
    rep = np.array([0, 0], dtype=np.int32) if p > 0 else None
    if rep is not None:
        a, b = xsplit.split_col_subsets(f, rep), xsplit.split_col_subsets(g, rep)
        for x, y in zip(a[0] + a[1], b[0] + b[1]):
            assert np.array_equal(x, y)
    assert collapse_identity(np.arange(p, dtype=np.int32), p) is None
    assert collapse_identity(None, p) is None
    if p > 1:
        assert collapse_identity(np.arange(p, dtype=np.int32)[::-1].copy(), p) is not None
        assert collapse_identity(np.arange(p - 1, dtype=np.int32), p) is not None


def test_device_row_index_checks_bounds():
    """Row ids handed to device gathers are range-checked on the host (numpy raises IndexError on
    the reference's host path; a device gather would read out of bounds silently)."""
    from tabmat_amd.util import device_row_index

    kind, idx = device_row_index(np.array([0, -1, 3]), 5)
    assert kind == "index" and idx.tolist() == [0, 4, 3]
    assert device_row_index(slice(1, 4), 5) == ("slice", 1, 4)
    for bad in ([5], [-6], [0, 7, 1]):
        with pytest.raises(IndexError):
            device_row_index(np.array(bad), 5)
    kind, idx = device_row_index(np.array([True, False, True, False, False]), 5)
    assert idx.tolist() == [0, 2]


def test_same_float_rejects_integer_operands():
    """A raw pointer to an integer d / v must never reach a tm_*_f32 / _f64 entry point."""
    from tabmat_amd import _device as D

    x = torch.zeros(4, dtype=torch.float64)
    D.same_float("op", x, torch.ones(4, dtype=torch.float64), None)
    with pytest.raises(TypeError):
        D.same_float("op", x, torch.ones(4, dtype=torch.int64))
    with pytest.raises(TypeError):
        D.same_float("op", x, torch.ones(4, dtype=torch.int32))
    with pytest.raises(TypeError):
        D.same_float("op", x, torch.ones(4, dtype=torch.float32))


def test_tuning_knobs_round_trip():
    v = C.c_int64(0)
    _lib.call("tm_tune_get", b"no_such_knob", 7, C.byref(v))
    assert v.value == 7
    _lib.call("tm_tune_set", b"unit_test_knob", 12)
    _lib.call("tm_tune_get", b"unit_test_knob", 7, C.byref(v))
    assert v.value == 12


@pytest.mark.parametrize("d12", [True, False])
@pytest.mark.parametrize("seed,m,dens", [(0, 128, 0.1), (1, 300, 0.12), (2, 513, 0.06), (3, 40, 0.9)])
def test_pair_block_list_covers_every_pair_once(seed, m, dens, d12):
    """CsrDev.pair_blocks (the static block list of tm_sparse_sandwich_blocks_*): replaying the list
    with the kernel's rules (8 x 8 blocks, b <= a triangle on diagonal tiles by COLUMN order, mirror)
    must give A' diag(d) A exactly once per pair -- rows with more than 8 and more than 16 entries
    in a 128-column chunk included."""
    from tabmat_amd.ext._types import CsrDev

    rng = np.random.default_rng(seed)
    n = 257
    A = sps.random(n, m, density=dens, format="csr", random_state=rng, dtype=np.float64)
    A.sort_indices()
    csr = CsrDev(torch.from_numpy(A.data.copy()), torch.from_numpy(A.indices.astype(np.int32)),
                 torch.from_numpy(A.indptr.astype(np.int64)), n, m)
    cm_data, _, cptr = csr.chunk_major()
    cm_ind = csr.chunk_cols32()           # block columns (the twin itself keeps one byte per entry)
    blocks, wg_tab, max_nb = csr.pair_blocks(n_wg=24, d12=d12)
    assert int(blocks.shape[1]) == (3 if d12 else 4)       # 12-byte descriptors (round 6) / the 16-byte form
    blocks = csr.unpack_blocks(blocks)
    cm_data, cm_ind, blocks, wg_tab = cm_data.numpy(), cm_ind.numpy(), blocks.numpy(), wg_tab.numpy()
    assert (cptr.numpy()[:, 1:] - cptr.numpy()[:, :-1]).max() > 8
    d = rng.random(n)
    nch = -(-m // 128)
    out = np.zeros((nch * 128, nch * 128))
    covered = np.zeros(len(blocks), dtype=int)
    slots = {}
    for part, slot, lo, hi, full_end, wfull, row_first, row_last in wg_tab:
        assert slot < max_nb and (part, slot) not in slots
        slots[(part, slot)] = 1
        I = int((np.sqrt(8 * part + 1) - 1) / 2)
        J = part - I * (I + 1) // 2
        assert lo <= full_end <= hi and 0 <= wfull <= 16
        assert (wfull == 16) == (full_end == hi) and (wfull == 0) == (full_end == lo)
        for k in range(lo, hi):
            covered[k] += 1
            pa, pb, row, meta = blocks[k]
            na, nb = meta & 255, (meta >> 8) & 255
            rep_a, rep_b = bool(meta & (1 << 16)), bool(meta & (1 << 17))
            assert 1 <= na <= 8 and 1 <= nb <= 8 and row_first <= row <= row_last
            assert (k < full_end) == (na > 4 and nb > 4)
            assert rep_a == (na <= 4 < nb) and rep_b == (nb <= 4 < na)
            steps = 8 if k < full_end else 4
            for t in range(8):                      # the kernel's lanes and DPP steps
                ta, tb = (t & 3 if rep_a else t), (t & 3 if rep_b else t)
                for s_ in range(steps):
                    u = t ^ s_                       # lane whose B entry arrives
                    ub = (u & 3) if rep_b else u
                    if ta < na and ub < nb:
                        ca, cb = cm_ind[pa + ta], cm_ind[pb + ub]
                        assert ca // 128 == I and cb // 128 == J
                        if I != J or cb <= ca:
                            out[ca, cb] += d[row] * cm_data[pa + ta] * cm_data[pb + ub]
    assert (covered == 1).all()
    out = out[:m, :m]
    full = np.tril(out) + np.tril(out, -1).T
    want = (A.T.multiply(d)).dot(A).toarray()
    assert np.abs(full - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0)


def test_pairs_cost_model_picks_the_regimes_it_was_measured_in():
    """profiles/r5_k2_pairs.txt (2M rows): 2048 columns @ 1.25 % and 4096 @ 0.625 % take the pair-stream kernel; the
    BASELINE shape (512 @ 5 %), 2048 @ 5 % (the block list serves dense rows better) and 8192 @ 0.05 % (the direct
    kernel) do not."""
    from tabmat_amd.ext import sparse as xs

    class Fake:
        def __init__(self, n, m, dens):
            self.n, self.m = n, m
            self._nnz = int(n * m * dens)
            self.data = type("T", (), {"numel": lambda s_: self._nnz})()

    n = 2_000_000
    assert xs.pairs_sandwich_pays(Fake(n, 2048, 0.0125))
    assert xs.pairs_sandwich_pays(Fake(n, 4096, 0.00625))
    assert xs.pairs_sandwich_pays(Fake(n, 4096, 0.002))             # 2.9-3.0 ms against 3.57 ms direct
    assert not xs.pairs_sandwich_pays(Fake(n, 512, 0.05))
    assert xs.pairs_sandwich_pays(Fake(n, 1024, 0.0125))            # 0.56 ms against 1.24 ms chunked
    assert not xs.pairs_sandwich_pays(Fake(n, 1024, 0.05))          # 6.3 ms against 2.8 ms on the block list
    assert not xs.pairs_sandwich_pays(Fake(n, 2048, 0.05))
    assert not xs.pairs_sandwich_pays(Fake(n, 8192, 0.0005))
    # the reference's own 'sparse_wide' design (benchmark/generate_matrices.py): pair stream 1.63 ms, direct 9.7 ms;
    # the same width ten times sparser over ten times the rows: direct 1.9 ms, pair stream 2.2 ms
    assert xs.pairs_sandwich_pays(Fake(40_000, 10_000, 0.01))
    assert not xs.pairs_sandwich_pays(Fake(400_000, 10_000, 0.001))
    assert not xs.pairs_sandwich_pays(Fake(40_000, 20_000, 0.01))          # beyond 128 column chunks


def test_api_members_of_the_reference_classes_exist():
    """Every public method / property the reference's six classes define (matrix_base.py, dense_matrix.py,
    sparse_matrix.py, categorical_matrix.py, split_matrix.py, standardized_mat.py) exists here under the same name --
    the list below was taken from the reference's class bodies; the hot-path ones are tested against the oracle
    elsewhere, this pins the drop-in surface (round 6 added StandardizedMatrix.multiply / names / __repr__ and
    CategoricalMatrix.cat / unpack)."""
    import tabmat_amd as tm

    common = ["matvec", "transpose_matvec", "sandwich", "standardize", "toarray", "astype", "getcol", "multiply",
              "get_names", "set_names", "column_names", "term_names", "A", "__matmul__", "__rmatmul__",
              "_get_col_means", "_get_col_stds"]
    for cls in (tm.DenseMatrix, tm.SparseMatrix, tm.CategoricalMatrix, tm.SplitMatrix):
        for name in common + ([] if cls is tm.SplitMatrix else ["_cross_sandwich"]):    # (split_matrix.py has none)
            assert hasattr(cls, name), (cls.__name__, name)
    for name in ("recover_orig", "tocsr", "to_sparse_matrix", "cat", "unpack", "drop_first", "categories"):
        assert hasattr(tm.CategoricalMatrix, name) or name in ("drop_first", "categories"), name
    for name in ("matvec", "transpose_matvec", "sandwich", "unstandardize", "getcol", "toarray", "A", "astype",
                 "multiply", "get_names", "set_names", "column_names", "term_names", "__getitem__", "__matmul__",
                 "__rmatmul__", "__repr__"):
        assert hasattr(tm.StandardizedMatrix, name), name


def test_standardized_names_multiply_and_categorical_unpack_on_the_host():
    import warnings

    import tabmat_amd as tm

    rng = np.random.default_rng(0)
    X = rng.standard_normal((7, 3))
    dm = tm.DenseMatrix(X, column_names=["a", "b", "c"])
    shift, mult = rng.standard_normal(3), rng.random(3) + 0.5
    S = tm.StandardizedMatrix(dm, shift, mult)
    assert S.column_names == ["a", "b", "c"] and S.get_names("term") == dm.get_names("term")
    S.column_names = ["x", "y", "z"]
    assert dm.column_names == ["x", "y", "z"]
    w = rng.random(7)
    np.testing.assert_allclose(S.multiply(w).toarray(), (X * mult + shift) * w[:, None])
    assert "StandardizedMat" in repr(S)
    codes = np.array([2, 0, 1, 1, 2])
    cm = tm.CategoricalMatrix(codes, categories=np.array(["u", "v", "w"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        cat = cm.unpack()
    assert list(cat) == ["w", "u", "v", "v", "w"]
