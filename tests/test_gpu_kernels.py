"""-m gpu parity tests, part 2: every HIP kernel against the CPU oracle on seeded random
inputs at sizes the oracle finishes in seconds.  Bars (BASELINE.json north_star): categorical
counts bit-exact; fp64 sandwich <= 1e-10 relative; fp32 paths are compared with the fp64
oracle at a tolerance that covers fp32 accumulation of n terms (stated per test)."""
import os

import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from _gpu_util import cross_err, nat_err, rel_err, to_tm_block, to_tm_split

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10


def _orc():
    from oracle import oracle as orc

    return orc


def _rows_subset(rng, n, frac=0.6):
    return np.sort(rng.choice(n, size=int(n * frac), replace=False)).astype(np.int32)


# ------------------------------------------------------------------ K1 dense sandwich
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (1000, 17), (4097, 64), (20000, 128), (3001, 100),
                                 (2500, 200), (1200, 300), (900, 700)])
def test_dense_sandwich_f64(order, n, k):
    import tabmat_amd as tm

    rng = np.random.default_rng(n * 31 + k)
    X = rng.standard_normal((n, k))
    X = np.asfortranarray(X) if order == "F" else X
    d = rng.random(n)
    res = tm.DenseMatrix(X).sandwich(d)
    ref = _orc().dense_sandwich(X, d, None, None)
    assert nat_err(res, ref) < F64_TOL
    assert np.array_equal(res, res.T)


@pytest.mark.parametrize("order", ["C", "F"])
def test_dense_sandwich_rows_cols(order):
    """rows = the nonzero-d subset, cols = a random subset (tests/test_fast_sandwich.py:51-98)."""
    import tabmat_amd as tm

    rng = np.random.default_rng(11)
    for n, k in [(777, 45), (5000, 150), (2000, 400)]:
        X = rng.standard_normal((n, k))
        X = np.asfortranarray(X) if order == "F" else X
        d = rng.random(n)
        d[rng.choice(n, size=n // 3, replace=False)] = 0.0
        rows = np.where(np.abs(d) > 1e-14)[0].astype(np.int32)
        cols = rng.choice(k, size=max(1, k // 2), replace=False).astype(np.int32)
        res = tm.DenseMatrix(X).sandwich(d, rows, cols)
        ref = _orc().dense_sandwich(X, d, rows, cols)
        assert nat_err(res, ref) < F64_TOL
        Xs = X[:, cols]
        assert nat_err(res, (Xs.T * d) @ Xs) < F64_TOL


@pytest.mark.parametrize("n,k", [(3000, 32), (50000, 256), (1500, 300)])
def test_dense_sandwich_f32(n, k):
    """fp32 block: compared with the fp64 oracle on the same fp32 data.  Tolerance 2e-5 relative
    to max|out| covers fp32 products accumulated over <= 5e4 rows (sqrt(n) * eps_f32 ~ 3e-5)."""
    import tabmat_amd as tm

    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, k)).astype(np.float32)
    d = rng.random(n).astype(np.float32)
    res = tm.DenseMatrix(X).sandwich(d)
    assert res.dtype == np.float32
    ref = _orc().dense_sandwich(X.astype(np.float64), d.astype(np.float64), None, None)
    assert nat_err(res, ref) < 2e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("k", list(range(1, 12)))
def test_dense_sandwich_narrow_widths(dtype, order, k):
    """1 .. 11 columns (csrc/syrk_narrow.hip; the reference's 'dense' design is 4M x 10): C-ordered blocks
    are staged through LDS with flat 16-byte loads, F-ordered ones read directly; row counts around the
    256-row tile and the 4-row alignment of a workgroup's slab."""
    import tabmat_amd as tm

    from tabmat_amd._lib import call

    rng = np.random.default_rng(7000 + k)
    tol = F64_TOL if dtype == np.float64 else 2e-5
    for n in (1, 2, 3, 5, 255, 256, 257, 1023, 5000, 70_001, 600_013):
        call("tm_tune_set", b"syrk_narrow_staged", n % 2)       # both forms at every width (default: by width)
        X = rng.standard_normal((n, k)).astype(dtype)
        X = np.asfortranarray(X) if order == "F" else X
        d = rng.random(n).astype(dtype)
        res = tm.DenseMatrix(X).sandwich(d)
        X64 = X.astype(np.float64)
        ref = X64.T @ (X64 * d.astype(np.float64)[:, None])
        assert res.shape == (k, k) and nat_err(res, ref) < tol, (n, k)
        assert np.array_equal(res, res.T)
    call("tm_tune_set", b"syrk_narrow_staged", -1)


# ------------------------------------------------------------------ K5 dense matvec
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_matvec_rmatvec(order, dtype):
    import tabmat_amd as tm

    rng = np.random.default_rng(3)
    n, k = 10007, 77
    X = rng.standard_normal((n, k)).astype(dtype)
    X = np.asfortranarray(X) if order == "F" else X
    mat = tm.DenseMatrix(X)
    v, w = rng.standard_normal(k).astype(dtype), rng.standard_normal(n).astype(dtype)
    rows, cols = _rows_subset(rng, n), np.sort(rng.choice(k, 30, replace=False)).astype(np.int32)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    X64 = X.astype(np.float64)
    assert rel_err(mat.matvec(v), X64 @ v) < tol
    assert rel_err(mat.matvec(v, cols), X64[:, cols] @ v[cols].astype(np.float64)) < tol
    assert rel_err(mat.transpose_matvec(w), X64.T @ w) < tol
    assert rel_err(mat.transpose_matvec(w, rows, cols),
                   X64[np.ix_(rows, cols)].T @ w[rows].astype(np.float64)) < tol
    orc = _orc()
    assert rel_err(mat.transpose_matvec(w, rows, cols), orc.dense_rmatvec(X, w, rows, cols)) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k", [1, 2, 3, 5, 7, 10, 12, 15, 18, 20, 31, 33, 63, 100, 127, 129, 200,
                               333, 500, 1000, 1279, 1280, 1281])
def test_dense_matvec_any_width(dtype, k):
    """Row lengths the streaming kernels have no lane split for: matvec stages whole rows in LDS
    (dense.hip dense_matvec_c_tile_kernel, any m <= 1280), transpose_matvec walks the slab as a
    flat vector array (dense_rmatvec_c_flat_kernel, column period <= 256).  Row counts around the tile and
    slab sizes, accumulation into out, a NaN in the last row staying in the last row."""
    import tabmat_amd as tm

    rng = np.random.default_rng(900 + k)
    tol = F64_TOL if dtype == np.float64 else 2e-4
    for n in (1, 2, 3, 5, 255, 256, 257, 1000, 4099, 70001):
        if n * k > 3e7:
            continue
        X = rng.standard_normal((n, k)).astype(dtype)
        v = rng.standard_normal(k).astype(dtype)
        mat = tm.DenseMatrix(X)
        ref = X.astype(np.float64) @ v.astype(np.float64)
        scale_mv = max(1.0, np.abs(ref).max())
        assert np.abs(mat.matvec(v) - ref).max() / scale_mv < tol, (n, k)
        out = np.full(n, 2.5, dtype=dtype)
        res = mat.matvec(v, out=out)
        assert res is out and np.abs(out - (ref + 2.5)).max() / scale_mv < tol
        w = rng.standard_normal(n).astype(dtype)
        ref_t = X.astype(np.float64).T @ w.astype(np.float64)
        scale = max(1.0, np.abs(ref_t).max())
        assert np.abs(mat.transpose_matvec(w) - ref_t).max() / scale < tol, (n, k)
        out_t = np.full(k, -1.5, dtype=dtype)
        mat.transpose_matvec(w, out=out_t)
        assert np.abs(out_t - (ref_t - 1.5)).max() / scale < tol
    X = rng.standard_normal((777, k)).astype(dtype)
    X[-1, -1] = np.nan
    got = tm.DenseMatrix(X).matvec(np.ones(k, dtype=dtype))
    assert np.isnan(got[-1]) and np.isfinite(got[:-1]).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k", [4, 16, 32, 64, 128, 256, 512, 520])
def test_dense_matvec_stream_path(dtype, k):
    """Unrestricted C-order matvec takes the 16-byte streaming kernel when a row is 8..128
    16-byte vectors long (dense.hip dense_matvec_c_stream_kernel); every lanes-per-row
    instantiation, row counts that do not fill the last wave step, and accumulation into out."""
    import tabmat_amd as tm

    rng = np.random.default_rng(100 + k)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    for n in (1, 7, 63, 1000, 4099):
        X = rng.standard_normal((n, k)).astype(dtype)
        v = rng.standard_normal(k).astype(dtype)
        mat = tm.DenseMatrix(X)
        ref = X.astype(np.float64) @ v.astype(np.float64)
        assert rel_err(mat.matvec(v), ref) < tol
        assert rel_err(mat.matvec(v), _orc().dense_matvec(X, v, None, None)) < tol
        out = np.full(n, 2.5, dtype=dtype)
        res = mat.matvec(v, out=out)
        assert res is out and rel_err(out, ref + 2.5) < tol
        # transpose_matvec and the weighted column second moments (K7) share the 16-byte
        # streaming kernel dense_rmatvec_c_stream_kernel
        w = rng.standard_normal(n).astype(dtype)
        ref_t = X.astype(np.float64).T @ w.astype(np.float64)
        scale = max(1.0, np.abs(ref_t).max())
        assert np.abs(mat.transpose_matvec(w) - ref_t).max() / scale < tol
        wts = rng.random(n).astype(dtype)
        wts /= wts.sum()
        means = (X.astype(np.float64) * wts[:, None].astype(np.float64)).sum(axis=0)
        ref_sd = np.sqrt((((X.astype(np.float64) - means) ** 2) * wts[:, None]).sum(axis=0))
        got_sd = mat._get_col_stds(wts, means.astype(dtype))
        assert np.abs(got_sd - ref_sd).max() < (1e-9 if dtype == np.float64 else 2e-3)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k", [(1000, 16), (4100, 100), (20000, 128), (16388, 7), (8192, 64)])
def test_dense_f_order_stream_paths(dtype, n, k):
    """F-ordered dense blocks (what from_csc / pandas hand over) with n a multiple of the
    16-byte vector: column-major LDS staging in the syrk (LOAD_F_VEC) and the LDS-staged-v
    transpose_matvec / K7 kernel (dense_rmatvec_f_stream_kernel), against the C-ordered result."""
    import tabmat_amd as tm

    rng = np.random.default_rng(n + k)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    X = rng.standard_normal((n, k)).astype(dtype)
    XF = np.asfortranarray(X)
    d = rng.random(n).astype(dtype)
    w = rng.standard_normal(n).astype(dtype)
    X64 = X.astype(np.float64)
    mf = tm.DenseMatrix(XF)
    ref = X64.T @ (d.astype(np.float64)[:, None] * X64)
    assert nat_err(mf.sandwich(d), ref) < tol
    assert nat_err(mf.sandwich(d), _orc().dense_sandwich(XF, d, None, None)) < tol
    ref_t = X64.T @ w.astype(np.float64)
    assert np.abs(mf.transpose_matvec(w) - ref_t).max() / max(1.0, np.abs(ref_t).max()) < tol
    wts = rng.random(n).astype(dtype)
    wts /= wts.sum()
    means = (X64 * wts[:, None].astype(np.float64)).sum(axis=0)
    ref_sd = np.sqrt((((X64 - means) ** 2) * wts[:, None]).sum(axis=0))
    assert np.abs(mf._get_col_stds(wts, means.astype(dtype)) - ref_sd).max() < (1e-9 if dtype == np.float64 else 2e-3)


# ------------------------------------------------------------------ K2 sparse sandwich
@pytest.mark.parametrize("idx_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("n,m,dens", [(200, 50, 0.05), (5000, 130, 0.05), (20000, 512, 0.05),
                                      (3000, 300, 0.3), (1000, 7, 0.9), (8000, 256, 0.09),
                                      (4001, 384, 0.06)])
def test_sparse_sandwich(idx_dtype, n, m, dens):
    import tabmat_amd as tm

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng)
    S = sps.csc_matrix((S.data, S.indices.astype(idx_dtype), S.indptr.astype(idx_dtype)),
                       shape=S.shape)
    d = rng.random(n)
    mat = tm.SparseMatrix(S)
    ref = _orc().sparse_sandwich(S, S.tocsr(), d, None, None)
    res = mat.sandwich(d)
    assert nat_err(res, ref) < F64_TOL
    assert np.array_equal(res, res.T)
    rows = _rows_subset(rng, n)
    cols = np.sort(rng.choice(m, size=max(1, m // 2), replace=False)).astype(np.int32)
    ref = _orc().sparse_sandwich(S, S.tocsr(), d, rows, cols)
    assert nat_err(mat.sandwich(d, rows, cols), ref) < F64_TOL


def test_sparse_sandwich_reference_seeds():
    """tests/test_fast_sandwich.py:12-31 (both dtypes; atol sqrt(eps) as in the reference)."""
    import tabmat_amd as tm

    for dtype in (np.float64, np.float32):
        np.random.seed(123)
        for _ in range(10):
            nrows, ncols = np.random.randint(200, size=2)
            A = cs.simulate_matrix(shape=(nrows, ncols), seed=None, dtype=dtype).tocsc()
            d = np.random.rand(A.shape[0]).astype(dtype)
            true = (A.T.multiply(d)).dot(A).toarray()
            if ncols == 0 or nrows == 0:
                continue
            out = tm.SparseMatrix(A).sandwich(d)
            np.testing.assert_allclose(true, out, atol=np.sqrt(np.finfo(dtype).eps))


# ------------------------------------------------------------------ K3 sparse x dense
@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n,m,r", [(100, 2, 4), (5000, 64, 33), (20000, 512, 128), (3000, 700, 20)])
def test_csr_dense_sandwich(order, n, m, r):
    import tabmat_amd as tm

    rng = np.random.default_rng(n + m + r)
    S = sps.random(n, m, density=0.05, format="csc", random_state=rng)
    B = rng.standard_normal((n, r))
    B = np.asfortranarray(B) if order == "F" else B
    d = rng.random(n)
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    orc = _orc()
    ref = orc.csr_dense_sandwich(S.tocsr(), B, d, None, None, None)
    assert cross_err(sm._cross_sandwich(dm, d, None), ref, d, S, B) < F64_TOL
    assert cross_err(dm._cross_sandwich(sm, d, None), ref.T, d, B, S) < F64_TOL
    rows = _rows_subset(rng, n)
    Ac = np.sort(rng.choice(m, size=max(1, m // 2), replace=False)).astype(np.int32)
    Bc = np.sort(rng.choice(r, size=max(1, r // 2), replace=False)).astype(np.int32)
    ref = orc.csr_dense_sandwich(S.tocsr(), B, d, rows, Ac, Bc)
    assert cross_err(sm._cross_sandwich(dm, d, rows, Ac, Bc), ref, d, S, B, rows, Ac, Bc) < F64_TOL


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,r,dens", [(1000, 50, 128, 0.05), (777, 37, 100, 0.2), (5000, 300, 256, 0.03),
                                        (130, 16, 72, 0.5), (64, 3, 68, 1.0), (4099, 513, 132, 0.01),
                                        (63, 1, 260, 0.9), (20000, 40, 66, 0.0)])
def test_csr_dense_sandwich_wide_ell(dtype, n, m, r, dens):
    """Dense operands with more than 64 columns take the wide interleaved-ELL kernel
    (tm_csr_dense_sandwich_ellw_*, sparse.hip K3 wide): part widths that are not multiples of 128,
    more than 256 sparse columns (blockIdx.z), row counts around the 64-row slab, blocks longer than
    the 3 prefetched chunks (dense columns), rows excluded by d == 0, an empty matrix."""
    import tabmat_amd as tm

    rng = np.random.default_rng(n + 3 * m + r)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng).astype(dtype)
    B = rng.standard_normal((n, r)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[::7] = 0
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    assert sm._ell(wide=True).wide
    ref = _orc().csr_dense_sandwich(S.tocsr().astype(np.float64), B.astype(np.float64),
                                    d.astype(np.float64), None, None, None)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    assert cross_err(sm._cross_sandwich(dm, d, None), ref, d, S, B) < tol
    rows = _rows_subset(rng, n)
    ref_r = _orc().csr_dense_sandwich(S.tocsr().astype(np.float64), B.astype(np.float64),
                                      d.astype(np.float64), rows, None, None)
    assert cross_err(sm._cross_sandwich(dm, d, rows), ref_r, d, S, B, rows) < tol


def test_very_sparse_block_keeps_the_compact_stream():
    """Every non-empty (slab, column group) of the interleaved-ELL twin costs 64 slots; a block
    with far less than one nonzero per slab and group falls back to the compact slab stream
    (SparseMatrix._ell -> None) instead of a twin dozens of times its size."""
    import tabmat_amd as tm

    rng = np.random.default_rng(77)
    n, m, r = 400_000, 512, 128
    S = sps.random(n, m, density=0.0005, format="csc", random_state=rng)
    B = rng.standard_normal((n, r))
    d = rng.random(n)
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    assert sm._ell(wide=True) is None
    ref = _orc().csr_dense_sandwich(S.tocsr(), B, d, None, None, None)
    assert cross_err(sm._cross_sandwich(dm, d, None), ref, d, S, B) < F64_TOL


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_f_ordered_dense_block_uses_row_major_twin(dtype):
    """An F-ordered dense block gets a row-major twin in HBM for the sandwich kernels
    (DenseMatrix._dev_c): same results as the column-major kernel variants and as the oracle;
    matvec / transpose_matvec keep the caller's layout."""
    import tabmat_amd as tm
    import tabmat_amd.dense_matrix as dmod

    specs, idx = cs.mixed_specs(9000, 130, 70, (9, 4), seed=5, dtype=dtype, order="F")
    assert specs[0][1].flags["F_CONTIGUOUS"]
    rng = np.random.default_rng(8)
    d = rng.random(9000).astype(dtype)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    ref = _orc().split_sandwich([cs.to_oracle_block(s) for s in specs], idx, d.astype(np.float64), None)
    assert dmod.ROW_MAJOR_TWIN
    mat = to_tm_split(specs, idx, dtype)
    res_twin = mat.sandwich(d)
    dense = mat.matrices[0]
    assert dense._dev().order_f == 1 and dense._dev_c().order_f == 0
    assert dense._dev_c() is dense._dev_c()          # built once
    assert rel_err(res_twin, ref) < tol and nat_err(res_twin, ref) < tol
    dmod.ROW_MAJOR_TWIN = False
    try:
        mat2 = to_tm_split(specs, idx, dtype)
        assert mat2.matrices[0]._dev_c().order_f == 1
        assert nat_err(mat2.sandwich(d), ref) < tol
    finally:
        dmod.ROW_MAJOR_TWIN = True
    v = rng.standard_normal(mat.shape[1]).astype(dtype)
    E = np.hstack([np.asarray(cs.spec_toarray(s), dtype=np.float64) for s in specs])
    assert rel_err(mat.matvec(v), E @ v.astype(np.float64)) < tol


# ------------------------------------------------------------------ K6 sparse matvec
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_matvec_rmatvec(dtype):
    import tabmat_amd as tm

    rng = np.random.default_rng(9)
    n, m = 30011, 213
    S = sps.random(n, m, density=0.04, format="csc", random_state=rng).astype(dtype)
    mat = tm.SparseMatrix(S)
    v, w = rng.standard_normal(m).astype(dtype), rng.standard_normal(n).astype(dtype)
    rows, cols = _rows_subset(rng, n), np.sort(rng.choice(m, 100, replace=False)).astype(np.int32)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    S64 = S.astype(np.float64)
    assert rel_err(mat.matvec(v), S64 @ v.astype(np.float64)) < tol
    assert rel_err(mat.matvec(v, cols), S64[:, cols] @ v[cols].astype(np.float64)) < tol
    assert rel_err(mat.transpose_matvec(w), S64.T @ w.astype(np.float64)) < tol
    assert rel_err(mat.transpose_matvec(w, rows, cols),
                   S64[rows][:, cols].T @ w[rows].astype(np.float64)) < tol
    orc = _orc()
    assert rel_err(mat.transpose_matvec(w, rows, cols), orc.csc_rmatvec(S, w, rows, cols)) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(1, 5, 1.0), (63, 40, 0.3), (64, 40, 0.3), (65, 40, 0.3),
                                      (300, 3000, 0.5), (20011, 512, 0.05), (5000, 64, 0.0),
                                      (4000, 10_000, 0.01), (1500, 15_000, 0.004), (1200, 20_000, 0.003)])
def test_sparse_matvec_stream_path(dtype, n, m, dens):
    """Unrestricted CSR matvec / transpose_matvec take the streaming kernels (sparse.hip K6 fast
    paths): row counts around the 64-row wave chunk, rows longer than the 1024-entry staging
    buffer (300 x 3000 at 50 %), an empty matrix, accumulation into out; short and wide blocks (the
    reference's 'sparse_wide' design is 40k x 10k: accumulators up to 128 KB of LDS, beyond that the
    generic kernel with a workgroup per 256 rows)."""
    import tabmat_amd as tm

    rng = np.random.default_rng(n * 7 + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng).astype(dtype)
    mat = tm.SparseMatrix(S)
    v, w = rng.standard_normal(m).astype(dtype), rng.standard_normal(n).astype(dtype)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    S64 = S.astype(np.float64)
    ref_mv, ref_tmv = S64 @ v.astype(np.float64), S64.T @ w.astype(np.float64)
    scale_mv, scale_tmv = max(1.0, np.abs(ref_mv).max()), max(1.0, np.abs(ref_tmv).max())
    assert np.abs(mat.matvec(v) - ref_mv).max() / scale_mv < tol
    assert np.abs(mat.transpose_matvec(w) - ref_tmv).max() / scale_tmv < tol
    orc = _orc()
    assert np.abs(mat.matvec(v) - orc.csr_matvec_unrestricted(S.tocsr(), v)).max() / scale_mv < tol
    out = np.full(m, -1.25, dtype=dtype)
    res = mat.transpose_matvec(w, out=out)
    assert res is out and np.abs(out - (ref_tmv - 1.25)).max() / scale_tmv < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("drop_first,missing", [(False, False), (True, False), (False, True), (True, True)])
def test_categorical_matvec_quads_and_fresh_output(dtype, drop_first, missing):
    """CategoricalMatrix.matvec (cat.hip cat_matvec_quad_kernel): four rows per lane, row counts around a quad and
    around the grid stride; without `out` the result is written into fresh storage (tm_cat_matvec_assign_*), with
    `out` it is added; column selections; bit-exact (one term per row)."""
    import tabmat_amd as tm

    rng = np.random.default_rng(4242)
    for n in (1, 2, 3, 4, 5, 7, 1023, 1024, 100_003, 2_100_001, 1_300_002):
        # (7000 / 12000 levels over >= 64 rows per level: the coefficient vector is staged in LDS)
        ncat = 7000 if n == 2_100_001 else 12_000 if n == 1_300_002 else int(rng.choice([1, 3, 50, 7000]))
        codes = rng.integers(0, ncat, n)
        if missing:
            codes = np.where(rng.random(n) < 0.2, -1, codes)
        mat = tm.CategoricalMatrix(codes, categories=np.arange(ncat), drop_first=drop_first, dtype=dtype,
                                   cat_missing_method="zero" if missing else "fail")
        k = mat.shape[1]
        v = rng.standard_normal(k).astype(dtype)
        col = codes - int(drop_first)
        ref = np.where(col >= 0, v[np.clip(col, 0, max(k - 1, 0))] if k else 0.0, 0.0).astype(dtype)
        got = mat.matvec(v)
        assert got.dtype == dtype and np.array_equal(got, ref), (n, ncat)
        out = np.full(n, 1.5, dtype=dtype)
        res = mat.matvec(v, out=out)
        assert res is out and np.array_equal(out, (ref + dtype(1.5)).astype(dtype))
        if k >= 2:
            cols = np.sort(rng.choice(k, size=max(1, k // 2), replace=False))
            keep = np.isin(col, cols)
            assert np.array_equal(mat.matvec(v, cols=cols), np.where(keep, ref, 0).astype(dtype))
        assert np.array_equal(mat.matvec(v, cols=np.array([], dtype=np.int64)), np.zeros(n, dtype=dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(60_000, 600, 0.05), (300_001, 37, 0.12), (40_000, 3000, 0.01)])
def test_sparse_matvec_16_bit_column_twin(dtype, n, m, dens, monkeypatch):
    """Blocks of a million entries and more stream a 16-bit twin of their column indices in the unrestricted matvec /
    transpose_matvec (tm_csr_{matvec,rmatvec}_u16_*): the same numbers as the int32 kernels (matvec bit for bit),
    accumulation into out, the restricted forms untouched."""
    import tabmat_amd as tm
    from tabmat_amd import _lib
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng).astype(dtype)
    assert S.nnz >= xs.CSR_U16_MIN_NNZ
    v, w = rng.standard_normal(m).astype(dtype), rng.standard_normal(n).astype(dtype)
    seen = []
    real = _lib.call
    monkeypatch.setattr("tabmat_amd.ext.sparse.call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    res = {}
    for flag in (False, True):
        monkeypatch.setattr(xs, "CSR_U16", flag)
        mat = tm.SparseMatrix(S)
        out = np.full(n, 0.5, dtype=dtype)
        res[flag] = (mat.matvec(v), mat.transpose_matvec(w), mat.matvec(v, out=out).copy(),
                     mat.matvec(v, cols=np.arange(0, m, 2)), mat.transpose_matvec(w, rows=np.arange(0, n, 3)))
    assert any(s_.startswith("tm_csr_matvec_u16_") for s_ in seen) and any(s_.startswith("tm_csr_rmatvec_u16_") for s_ in seen)
    tol = F64_TOL if dtype == np.float64 else 1e-4
    S64 = S.astype(np.float64)
    ref_mv, ref_t = S64 @ v.astype(np.float64), S64.T @ w.astype(np.float64)
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][2], res[False][2])
    assert np.abs(res[True][0] - ref_mv).max() / max(1.0, np.abs(ref_mv).max()) < tol
    assert np.abs(res[True][1] - ref_t).max() / max(1.0, np.abs(ref_t).max()) < tol
    assert np.array_equal(res[True][3], res[False][3])
    assert np.abs(res[True][4] - res[False][4]).max() / max(1.0, np.abs(ref_t).max()) < tol


# ------------------------------------------------------------------ K4 categorical family
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ncat", [3, 1000, 10_000, 58_059])
@pytest.mark.parametrize("drop_first,missing", [(False, False), (True, False), (False, True),
                                                (True, True)])
def test_categorical_counts_bit_exact(dtype, ncat, drop_first, missing):
    """d == 1: sandwich diagonal and transpose_matvec are exact integer counts -> bit-exact."""
    import tabmat_amd as tm

    rng = np.random.default_rng(ncat)
    n = 400_000
    codes = rng.integers(0, ncat, n).astype(np.int32)
    if missing:
        codes[rng.random(n) < 0.03] = -1
    mat = to_tm_block(("cat", codes, ncat, drop_first), dtype)
    ones = np.ones(n, dtype=dtype)
    counts = np.bincount(codes[codes >= 0], minlength=ncat)[int(drop_first):].astype(dtype)
    diag = mat.sandwich(ones).diagonal()
    assert diag.dtype == dtype and np.array_equal(diag, counts)
    assert np.array_equal(mat.transpose_matvec(ones), counts)
    rows = _rows_subset(rng, n, 0.5)
    c2 = codes[rows]
    counts_r = np.bincount(c2[c2 >= 0], minlength=ncat)[int(drop_first):].astype(dtype)
    assert np.array_equal(mat.sandwich(ones, rows).diagonal(), counts_r)
    orc = _orc()
    assert np.array_equal(counts_r, orc.sandwich_categorical(codes, ones, rows, mat.shape[1], drop_first))


@pytest.mark.parametrize("drop_first,missing", [(False, False), (True, True)])
def test_categorical_weighted(drop_first, missing):
    import tabmat_amd as tm

    rng = np.random.default_rng(2)
    n, ncat = 300_000, 5000
    codes = rng.integers(0, ncat, n).astype(np.int32)
    if missing:
        codes[rng.random(n) < 0.03] = -1
    mat = to_tm_block(("cat", codes, ncat, drop_first))
    A = cs.spec_toarray(("cat", codes[:2000], ncat, drop_first))
    d = rng.random(n)
    orc = _orc()
    k = mat.shape[1]
    ref = orc.sandwich_categorical(codes, d, None, k, drop_first)
    assert rel_err(mat.sandwich(d).diagonal(), ref) < F64_TOL
    rows = _rows_subset(rng, n)
    cols = np.sort(rng.choice(k, 700, replace=False)).astype(np.int32)
    out = np.zeros(k)
    orc.cat_transpose_matvec(codes, d, k, rows, cols, out, drop_first)
    assert rel_err(mat.transpose_matvec(d, rows, cols), out[cols]) < F64_TOL
    # matvec (gather) is exact
    v = rng.standard_normal(k)
    ref = np.zeros(n)
    orc.cat_matvec(codes, v, n, None, k, ref, drop_first)
    assert np.array_equal(mat.matvec(v), ref)
    ref = np.zeros(n)
    orc.cat_matvec(codes, v, n, cols, k, ref, drop_first)
    assert np.array_equal(mat.matvec(v, cols), ref)
    del A


@pytest.mark.parametrize("ni,nj", [(3, 4), (256, 96), (1000, 300), (58_059, 27)])
@pytest.mark.parametrize("drops", [(False, False), (True, False), (True, True)])
def test_cat_cat(ni, nj, drops):
    import tabmat_amd as tm

    rng = np.random.default_rng(ni + nj)
    n = 200_000
    ci = rng.integers(0, ni, n).astype(np.int32)
    cj = rng.integers(0, nj, n).astype(np.int32)
    cj[rng.random(n) < 0.02] = -1
    mi = to_tm_block(("cat", ci, ni, drops[0]))
    mj = to_tm_block(("cat", cj, nj, drops[1]))
    d = rng.random(n)
    rows = _rows_subset(rng, n)
    orc = _orc()
    for r in (None, rows):
        ref = orc.sandwich_cat_cat(ci, cj, mi.shape[1], mj.shape[1], d, r, drops[0], drops[1])
        assert cross_err(mi._cross_sandwich(mj, d, r), ref, d, ("cat", ci, mi.shape[1], drops[0]),
                         ("cat", cj, mj.shape[1], drops[1]), r) < F64_TOL
    ones = np.ones(n)
    ref = orc.sandwich_cat_cat(ci, cj, mi.shape[1], mj.shape[1], ones, None, drops[0], drops[1])
    assert np.array_equal(mi._cross_sandwich(mj, ones, None), ref)  # counts: bit-exact


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ni,nj", [(1000, 1000), (40, 5000), (7000, 33)])
def test_cat_cat_level_sorted(dtype, ni, nj):
    """Tables of several LDS tiles take the level-sorted kernel (tm_cat_cat_sandwich_sorted_*, static twin of the
    pair): skewed levels (one hot cell), missing codes on both sides, empty levels, drop_first, a row restriction,
    column selections, counts bit-exact; the twin is reused across calls and not confused between partners."""
    import tabmat_amd as tm
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(ni * 3 + nj)
    n = 520_000
    assert xsplit.cat_cat_sorted_pays(n, ni - 1, nj)
    ci = np.minimum((rng.pareto(1.0, n) * 3).astype(np.int64), ni - 1).astype(np.int32)      # level 0 is hot
    cj = rng.integers(0, nj, n).astype(np.int32)
    cj[ci == 0] = 1                                                                          # ... in ONE cell
    ci[rng.random(n) < 0.03] = -1
    cj[rng.random(n) < 0.03] = -1
    ci[ci == 5] = 6                                                                          # an empty level
    mi = tm.CategoricalMatrix(ci, categories=np.arange(ni), drop_first=True, dtype=dtype, cat_missing_method="zero")
    mj = tm.CategoricalMatrix(cj, categories=np.arange(nj), drop_first=False, dtype=dtype, cat_missing_method="zero")
    mk = tm.CategoricalMatrix((cj + 1) % nj, categories=np.arange(nj), dtype=dtype)          # another partner
    d = rng.random(n).astype(dtype)

    def ref_table(a, b, w, rows=None):
        t = np.zeros((ni - 1, nj))
        ok = (a >= 1) & (b >= 0)
        if rows is not None:
            m = np.zeros(n, dtype=bool)
            m[rows] = True
            ok &= m
        np.add.at(t, (a[ok] - 1, b[ok]), w[ok].astype(np.float64))
        return t

    tol = 1e-12 if dtype == np.float64 else 2e-5
    ref = ref_table(ci, cj, d)
    got = mi._cross_sandwich(mj, d)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= tol * max(1.0, ref.max())
    got2 = mi._cross_sandwich(mk, d)
    ref2 = ref_table(ci, (cj + 1) % nj, d)
    assert np.abs(got2 - ref2).max() <= tol * max(1.0, ref2.max())
    assert np.abs(mi._cross_sandwich(mj, d) - ref).max() <= tol * max(1.0, ref.max())          # twin reused
    rows = _rows_subset(rng, n)
    refr = ref_table(ci, cj, d, rows)
    lc = np.sort(rng.choice(ni - 1, size=min(20, ni - 1), replace=False)).astype(np.int32)
    rc = np.sort(rng.choice(nj, size=min(25, nj), replace=False)).astype(np.int32)
    gotr = mi._cross_sandwich(mj, d, rows, lc, rc)
    assert np.abs(gotr - refr[lc][:, rc]).max() <= tol * max(1.0, refr.max())
    ones = np.ones(n, dtype=dtype)
    assert np.array_equal(mi._cross_sandwich(mj, ones), ref_table(ci, cj, ones))             # counts: exact
    # the whole SplitMatrix product goes the same way
    sp = tm.SplitMatrix([mi, mj])
    S = sp.sandwich(d)
    assert np.abs(S[: ni - 1, ni - 1:] - ref).max() <= tol * max(1.0, ref.max())


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_many_level_categoricals_share_one_pass_over_a_narrow_dense_block(order, dtype, monkeypatch):
    """Categoricals with more levels than the fused groups take (> 496) against a dense block of a few columns: ONE
    launch of the LDS-tile kernel for all of them (the reference's design dense_cat), with a row restriction, a
    column selection and a second, wide dense block that keeps its own path."""
    import tabmat_amd as tm
    from tabmat_amd import _lib

    rng = np.random.default_rng(77)
    n = 30_000
    c1 = rng.integers(0, 600, n)
    c2 = np.where(rng.random(n) < 0.05, -1, rng.integers(0, 700, n))
    c3 = rng.integers(0, 12, n)
    Xn = rng.standard_normal((n, 5)).astype(dtype)
    Xn = np.asfortranarray(Xn) if order == "F" else Xn
    Xw = rng.standard_normal((n, 40)).astype(dtype)
    blocks = [tm.CategoricalMatrix(c1, categories=np.arange(600), dtype=dtype),
              tm.DenseMatrix(Xn),
              tm.CategoricalMatrix(c2, categories=np.arange(700), drop_first=True, dtype=dtype, cat_missing_method="zero"),
              tm.CategoricalMatrix(c3, categories=np.arange(12), dtype=dtype),
              tm.DenseMatrix(Xw)]
    X = tm.SplitMatrix(blocks)
    E = X.toarray().astype(np.float64)
    d = rng.random(n).astype(dtype)
    seen = []
    real = _lib.call
    monkeypatch.setattr("tabmat_amd.ext.split.call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    tol = F64_TOL if dtype == np.float64 else 2e-5
    ref = E.T @ (E * d.astype(np.float64)[:, None])
    assert nat_err(X.sandwich(d), ref) < tol
    # the 600- and the 700-level block against the 5 columns: one fused launch
    assert sum(s_.startswith("tm_multi_cat_dense_sandwich_") for s_ in seen) <= 3
    rows = _rows_subset(rng, n)
    cols = np.sort(rng.choice(E.shape[1], size=E.shape[1] // 2, replace=False))
    refr = E[rows][:, cols].T @ (E[rows][:, cols] * d.astype(np.float64)[rows, None])
    assert nat_err(X.sandwich(d, rows=rows, cols=cols), refr) < tol


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("ncat,k", [(5, 3), (256, 128), (1000, 40), (20_000, 16)])
def test_cat_dense(order, ncat, k):
    import tabmat_amd as tm

    rng = np.random.default_rng(ncat + k)
    n = 60_000
    codes = rng.integers(0, ncat, n).astype(np.int32)
    codes[rng.random(n) < 0.02] = -1
    X = rng.standard_normal((n, k))
    X = np.asfortranarray(X) if order == "F" else X
    cm, dm = to_tm_block(("cat", codes, ncat, True)), tm.DenseMatrix(X)
    d = rng.random(n)
    orc = _orc()
    ref = orc.sandwich_cat_dense(codes, cm.shape[1], d, X, None, None, True)
    cspec = ("cat", codes, cm.shape[1], True)
    assert cross_err(cm._cross_sandwich(dm, d), ref, d, cspec, X) < F64_TOL
    assert cross_err(dm._cross_sandwich(cm, d), ref.T, d, X, cspec) < F64_TOL
    rows = _rows_subset(rng, n)
    jc = np.sort(rng.choice(k, size=max(1, k // 2), replace=False)).astype(np.int32)
    lc = np.sort(rng.choice(cm.shape[1], size=max(1, cm.shape[1] // 3), replace=False)).astype(np.int32)
    ref = orc.sandwich_cat_dense(codes, cm.shape[1], d, X, rows, jc, True)[lc]
    assert cross_err(cm._cross_sandwich(dm, d, rows, lc, jc), ref, d, cspec, X, rows, lc, jc) < F64_TOL


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k,ncats", [(1, 4, (3,)), (31, 32, (5, 2)), (32, 36, (7,)), (33, 128, (256, 96, 32)),
                                       (4097, 132, (11, 3, 2, 9)), (70_000, 8, (300, 40)), (5000, 260, (64,))])
def test_multi_cat_dense_wide_kernel(dtype, n, k, ncats):
    """tm_multi_cat_dense_sandwich_* on its wide-load path (cat.hip multi_cat_dense_wide_kernel):
    1..4 categoricals, row counts around the 32-row wave step, column counts that are not multiples
    of the 16 * VEC part width, drop_first, missing codes, rows with d == 0 holding inf."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import split as xsplit

    rng = np.random.default_rng(n + k + len(ncats))
    X = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[::5] = 0
    X[::5, 0] = np.inf                      # excluded rows must not leak inf * 0
    dm = tm.DenseMatrix(X)
    cats, blocks = [], []
    for ci, nc in enumerate(ncats):
        codes = rng.integers(0, nc, n).astype(np.int32)
        codes[rng.random(n) < 0.05] = -1
        drop = bool(ci % 2)
        cm = to_tm_block(("cat", codes, nc, drop), dtype)
        blocks.append((codes, cm.shape[1], drop))
        cats.append((cm._dev(), cm.shape[1], drop))
    assert xsplit.multi_cat_dense_wide_ok(cats, dm._dev())
    res = D.to_host(xsplit.multi_cat_dense_sandwich(cats, D.to_dev(d), dm._dev()))
    Xc = X.astype(np.float64).copy()
    Xc[::5, 0] = 0.0
    orc, off = _orc(), 0
    tol = F64_TOL if dtype == np.float64 else 1e-4
    for codes, ncol, drop in blocks:
        ref = orc.sandwich_cat_dense(codes, ncol, d.astype(np.float64), Xc, None, None, drop)
        assert cross_err(res[off:off + ncol], ref, d, ("cat", codes, ncol, drop), Xc) < tol
        off += ncol
    assert off == res.shape[0]


@pytest.mark.parametrize("ncat,m", [(5, 3), (256, 512), (1000, 100), (3000, 2000)])
def test_cat_sparse(ncat, m):
    import tabmat_amd as tm

    rng = np.random.default_rng(ncat + m)
    n = 50_000
    codes = rng.integers(0, ncat, n).astype(np.int32)
    codes[rng.random(n) < 0.02] = -1
    S = sps.random(n, m, density=0.05, format="csc", random_state=rng)
    cm, sm = to_tm_block(("cat", codes, ncat, False)), tm.SparseMatrix(S)
    d = rng.random(n)
    orc = _orc()
    ref = orc.sandwich_cat_sparse(codes, ncat, d, S.tocsr(), None, None, None)
    cspec = ("cat", codes, ncat, False)
    assert cross_err(cm._cross_sandwich(sm, d), ref, d, cspec, S) < F64_TOL
    assert cross_err(sm._cross_sandwich(cm, d, None), ref.T, d, S, cspec) < F64_TOL
    rows = _rows_subset(rng, n)
    rc = np.sort(rng.choice(m, size=max(1, m // 2), replace=False)).astype(np.int32)
    lc = np.sort(rng.choice(ncat, size=max(1, ncat // 3), replace=False)).astype(np.int32)
    ref = orc.sandwich_cat_sparse(codes, ncat, d, S.tocsr(), rows, lc, rc)
    assert cross_err(cm._cross_sandwich(sm, d, rows, lc, rc), ref, d, cspec, S, rows, lc, rc) < F64_TOL
    # the reference computes this term with scipy.sparse (categorical_matrix.py:825-838)
    onehot = sps.csr_matrix((d[codes >= 0], (np.nonzero(codes >= 0)[0], codes[codes >= 0])),
                            shape=(n, ncat))
    assert cross_err(cm._cross_sandwich(sm, d), (onehot.T @ S.tocsr()).toarray(), d, cspec, S) < F64_TOL


# ------------------------------------------------------------------ SplitMatrix, cfg4 shape
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["C", "F"])
def test_split_mixed_cfg4_shape(dtype, order):
    """BASELINE.json config 4 at 40k rows: dense 128 + CSC 512 @5% + cats (256, 96, 32)."""
    specs, idx = cs.mixed_specs(40_000, 128, 512, (256, 96, 32), seed=3, dtype=dtype, order=order)
    mat = to_tm_split(specs, idx, dtype)
    blocks = [cs.to_oracle_block(s) for s in specs]
    orc = _orc()
    rng = np.random.default_rng(0)
    n, p = mat.shape
    assert p == 1024
    d = rng.random(n).astype(dtype)
    res = mat.sandwich(d)
    assert res.dtype == np.float64 and res.shape == (p, p)
    ref = orc.split_sandwich(blocks, idx, d)
    tol = F64_TOL if dtype == np.float64 else 5e-5
    assert rel_err(res, ref) < tol
    # entry by entry at the natural scale sqrt(S_ii S_jj): every block of the result, not just the largest
    assert nat_err(res, ref) < tol
    assert np.array_equal(res, res.T)
    rows = _rows_subset(rng, n)
    cols = np.sort(rng.choice(p, size=300, replace=False))
    ref_rc = orc.split_sandwich(blocks, idx, d, rows, cols)
    got_rc = mat.sandwich(d, rows, cols)
    assert rel_err(got_rc, ref_rc) < tol
    assert nat_err(got_rc, ref_rc) < tol
    v = rng.standard_normal(p).astype(dtype)
    w = rng.standard_normal(n).astype(dtype)
    mtol = F64_TOL if dtype == np.float64 else 1e-4
    assert rel_err(mat.matvec(v), orc.split_matvec(blocks, idx, v)) < mtol
    assert rel_err(mat.matvec(v, cols), orc.split_matvec(blocks, idx, v, cols)) < mtol
    assert rel_err(mat.transpose_matvec(w), orc.split_transpose_matvec(blocks, idx, w)) < mtol
    assert rel_err(mat.transpose_matvec(w, rows, cols),
                   orc.split_transpose_matvec(blocks, idx, w, rows, cols)) < mtol


def test_device_in_device_out():
    """torch cuda tensors in -> torch cuda tensors out (no host round trip)."""
    import torch

    specs, idx = cs.mixed_specs(5000, 16, 40, (7, 5), seed=1)
    mat = to_tm_split(specs, idx)
    rng = np.random.default_rng(0)
    d = rng.random(mat.shape[0])
    host = mat.sandwich(d)
    dev = mat.sandwich(torch.from_numpy(d).cuda())
    assert isinstance(dev, torch.Tensor) and dev.is_cuda
    # LDS atomics make the summation order run-dependent: equal to rounding, not bitwise
    np.testing.assert_allclose(dev.cpu().numpy(), host, rtol=1e-13, atol=1e-13)
    v = rng.random(mat.shape[1])
    np.testing.assert_allclose(mat.matvec(torch.from_numpy(v).cuda()).cpu().numpy(), mat.matvec(v),
                               rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(mat.transpose_matvec(torch.from_numpy(d).cuda()).cpu().numpy(),
                               mat.transpose_matvec(d), rtol=1e-13, atol=1e-12)


def test_standardized_split():
    """StandardizedMatrix on device blocks (tests/test_standardized_mat.py style)."""
    specs, idx = cs.mixed_specs(3000, 6, 9, (4, 3), seed=4)
    mat = to_tm_split(specs, idx)
    rng = np.random.default_rng(1)
    n, p = mat.shape
    w = rng.random(n)
    w /= w.sum()
    std, means, stds = mat.standardize(w, True, True)
    A = mat.toarray()
    np.testing.assert_allclose(means, A.T @ w, rtol=1e-10)
    np.testing.assert_allclose(stds, np.sqrt(((A - means) ** 2).T @ w), rtol=1e-8, atol=1e-12)
    S = std.toarray()
    d = rng.random(n)
    np.testing.assert_allclose(std.sandwich(d), (S.T * d) @ S, rtol=1e-8, atol=1e-9)
    v = rng.random(p)
    np.testing.assert_allclose(std.matvec(v), S @ v, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(std.transpose_matvec(d), S.T @ d, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_column_second_moments_k7(order, dtype):
    """transpose_square_dot_weights (ext/dense.pyx:103-122, ext/sparse.pyx:262-282) vs oracle."""
    import tabmat_amd as tm

    rng = np.random.default_rng(17)
    n, k = 20_011, 37
    X = rng.standard_normal((n, k)).astype(dtype)
    X = np.asfortranarray(X) if order == "F" else X
    S = sps.random(n, 90, density=0.05, format="csc", random_state=rng).astype(dtype)
    w = rng.random(n).astype(dtype)
    w /= w.sum()
    orc = _orc()
    dm, sm = tm.DenseMatrix(X), tm.SparseMatrix(S)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    means = dm.transpose_matvec(w)
    ref = np.sqrt(np.maximum(orc.dense_col_sq_dev(X, w, means.astype(dtype)), 0))
    assert rel_err(dm._get_col_stds(w, means), ref) < tol
    smeans = sm.transpose_matvec(w)
    ref = np.sqrt(np.maximum(orc.csc_col_sq(S, w) - smeans.astype(np.float64) ** 2, 0))
    assert rel_err(sm._get_col_stds(w, smeans), ref) < tol * 10


def test_row_restriction_ignores_non_finite_excluded_rows():
    """The fast paths implement `rows` as a masked d.  Rows that are NOT selected must not
    contribute even if they hold inf / nan (the reference never reads them)."""
    specs, idx = cs.mixed_specs(6000, 8, 40, (7, 5), seed=11)
    X = specs[0][1].copy()
    S = specs[1][1].copy().tolil()
    rng = np.random.default_rng(3)
    rows = np.sort(rng.choice(6000, 4000, replace=False)).astype(np.int32)
    excluded = np.setdiff1d(np.arange(6000), rows)
    X[excluded[:50], 2] = np.inf
    X[excluded[50:100], 5] = np.nan
    S[int(excluded[7]), 3] = np.inf
    specs = [("dense", X), ("sparse", sps.csc_matrix(S))] + list(specs[2:])
    mat = to_tm_split(specs, idx)
    d = rng.random(6000)
    res = mat.sandwich(d, rows=rows)
    assert np.isfinite(res).all()
    Xc, Sc = X.copy(), sps.lil_matrix(S)
    Xc[excluded] = 0.0
    Sc[int(excluded[7]), 3] = 0.0
    clean = [("dense", Xc), ("sparse", sps.csc_matrix(Sc))] + list(specs[2:])
    ref = _orc().split_sandwich([cs.to_oracle_block(s) for s in clean], idx, d, rows)
    assert nat_err(res, ref) < F64_TOL


@pytest.mark.skipif(os.environ.get("TABMAT_AMD_DETERMINISTIC", "0") not in ("", "0"),
                    reason="the fixed-order paths of TABMAT_AMD_DETERMINISTIC=1 synchronise with the host (not capturable)")
def test_sandwich_graph_replay_matches_eager():
    """SplitMatrix.sandwich_graph replays the captured launch sequence (tabmat_amd/graph.py):
    same result as the eager call for new d, with and without rows/cols, and after the library's
    workspace moved (re-capture on tm_workspace_generation change)."""
    import torch

    import tabmat_amd as tm

    specs, idx = cs.mixed_specs(20_000, 32, 48, (16, 9, 5), seed=5)
    X = to_tm_split(specs, idx)
    rng = np.random.default_rng(6)
    d0 = torch.from_numpy(rng.random(20_000)).cuda()
    f = X.sandwich_graph(d0)
    for seed in (1, 2):
        d = torch.from_numpy(np.random.default_rng(seed).random(20_000)).cuda()
        got = f(d).clone()
        ref = X.sandwich(d)
        assert nat_err(got.cpu().numpy(), ref.cpu().numpy()) < F64_TOL
    rows = _rows_subset(rng, 20_000)
    cols = np.sort(rng.choice(X.shape[1], 40, replace=False))
    fr = X.sandwich_graph(d0, rows, cols)
    assert nat_err(fr(d0).cpu().numpy(), X.sandwich(d0, rows, cols).cpu().numpy()) < F64_TOL
    # a larger product moves the workspace: the captured graphs must notice and re-capture
    big = tm.DenseMatrix(rng.standard_normal((300_000, 700)))
    big.sandwich(rng.random(300_000))
    d = torch.from_numpy(np.random.default_rng(9).random(20_000)).cuda()
    assert nat_err(f(d).cpu().numpy(), X.sandwich(d).cpu().numpy()) < F64_TOL


def test_sandwich_graph_replay_of_the_round_3_kernels():
    """The cfg4 geometry (dense 128 -> int8-sliced syrk with its device-side hand-over, sparse 512 -> pair-block
    K2 and the compact K3 stream) captured into a HIP graph: replays match the eager call, also for weights
    that leave the int8 envelope (the flag is evaluated on the device at replay time)."""
    import torch

    specs, idx = cs.mixed_specs(20_000, 128, 512, (32, 16, 8), seed=11)
    X = to_tm_split(specs, idx)
    rng = np.random.default_rng(12)
    d0 = torch.from_numpy(rng.random(20_000)).cuda()
    f = X.sandwich_graph(d0)
    blocks = [cs.to_oracle_block(s_) for s_ in specs]
    for seed, shift in ((1, 0.0), (2, 0.3), (3, 0.0)):          # 0.3: negative weights -> f64 kernel at replay
        dh = np.random.default_rng(seed).random(20_000) - shift
        got = f(torch.from_numpy(dh).cuda()).cpu().numpy()
        assert rel_err(got, _orc().split_sandwich(blocks, idx, dh)) < F64_TOL   # (weights of both signs: no natural scale)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(30_011, 2048, 0.0125), (20_000, 1024, 0.003), (9000, 700, 0.01),
                                      (5000, 384, 0.002)])
def test_sparse_sandwich_few_nonzeros_per_chunk(n, m, dens, dtype):
    """Wide and sparse blocks: the chunked K2 kernel runs with 4 or 2 slots per row and chunk
    (sparse.hip, template parameter S); a few dense rows exercise the overhang steps and the
    long-list fallback of those geometries."""
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="lil", random_state=rng, dtype=np.float64)
    for r in rng.choice(n, 40, replace=False):          # rows with 3 .. 40 nonzeros per chunk
        cols = rng.choice(m, int(m * rng.uniform(0.03, 0.3)), replace=False)
        S[r, cols] = rng.standard_normal(len(cols))
    S = sps.csc_matrix(S).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 10)] = 0
    sm = tm.SparseMatrix(S)
    got = sm.sandwich(d)
    want = orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d, None, None)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    rows = np.sort(rng.choice(n, n // 5, replace=False))
    got = sm.sandwich(d, rows=rows)
    want = orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d, rows, None)
    assert np.abs(got - want).max() <= tol * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,dens", [(20_000, 4096, 0.002), (3001, 1500, 0.004), (64, 1100, 0.05)])
def test_sparse_sandwich_direct(n, m, dens, dtype):
    """Wide, very sparse blocks: one L2 atomic per pair (sparse_direct.hip) against the oracle,
    with a few long rows (several 16-entry blocks per row) and zero weights."""
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="lil", random_state=rng, dtype=np.float64)
    for r in rng.choice(n, 12, replace=False):
        cols = rng.choice(m, int(rng.integers(17, 90)), replace=False)
        S[r, cols] = rng.standard_normal(len(cols))
    S = sps.csc_matrix(S).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 10)] = 0
    sm = tm.SparseMatrix(S)
    got = D.to_host(xs.sparse_sandwich_direct(sm._dev(), D.to_dev(d)))
    want = orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d, None, None)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    if xs.direct_sandwich_pays(sm._dev()):          # and through the public entry point
        rows = np.sort(rng.choice(n, n // 3, replace=False))
        cols = np.arange(0, m, 3)
        got = sm.sandwich(d, rows=rows, cols=cols)
        want = orc.sparse_sandwich(sps.csc_matrix(S), sps.csr_matrix(S), d, rows, cols)
        assert np.abs(got - want).max() <= tol * np.abs(want).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order", ["C", "F"])
def test_narrow_column_selection_dense_block_form(dtype, order):
    """A narrow `cols=` selection runs as [selected dense + sparse columns written out as one dense
    block | categorical blocks] (tm_csr_densify_cols_*, tm_dense_gather_cols_*): against the oracle,
    with rows, with selections that leave out whole blocks, and through StandardizedMatrix."""
    import tabmat_amd as tm
    import tabmat_amd.split_matrix as smod
    from oracle import oracle as orc

    n = 12_345
    specs, idx = cs.mixed_specs(n, 40, 90, (30, 7, 120), seed=5, dtype=dtype, order=order,
                                missing=True, drop_first=True)
    X = to_tm_split(specs, idx, dtype)
    blocks = [cs.to_oracle_block(s) for s in specs]
    p = X.shape[1]
    rng = np.random.default_rng(3)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 8)] = 0
    rows = np.sort(rng.choice(n, n // 3, replace=False))
    tol = 1e-10 if dtype == np.float64 else 3e-5
    sels = [np.sort(rng.choice(p, 17, replace=False)), np.arange(3, 40, 5), np.array([p - 1]),
            np.sort(rng.choice(p, p // 3, replace=False)), np.arange(130, p)]
    for cols in sels:
        for r in (None, rows):
            got = X.sandwich(d, rows=r, cols=cols)
            want = orc.split_sandwich(blocks, idx, d, r, cols)
            assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    # same numbers as the generic restricted kernels
    cols = sels[0]
    a = X.sandwich(d, cols=cols)
    old, smod.NARROW_COLS = smod.NARROW_COLS, 0
    try:
        b = X.sandwich(d, cols=cols)
    finally:
        smod.NARROW_COLS = old
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    # the standardized view (inner sandwich + column sums from the same pass)
    shift, mult = rng.standard_normal(p), rng.random(p) + 0.5
    E = orc.split_toarray(blocks, idx) * mult + shift
    for r in (None, rows):
        got = tm.StandardizedMatrix(X, shift, mult).sandwich(d, r, cols)
        Er = E[:, cols] if r is None else E[np.ix_(r, cols)]
        dr = d.astype(np.float64) if r is None else d[r].astype(np.float64)
        want = Er.T @ (dr[:, None] * Er)
        assert np.abs(np.asarray(got) - want).max() <= (1e-9 if dtype == np.float64 else 5e-3) * max(1.0, np.abs(want).max())


def test_matvec_column_selection_does_not_touch_excluded_inf_columns():
    """matvec with `cols` runs as X (v with zeros on the excluded columns) only when the blocks
    hold no inf / nan: an excluded column with an inf must not leak 0 x inf = nan."""
    import tabmat_amd as tm

    rng = np.random.default_rng(0)
    n = 5000
    A = rng.standard_normal((n, 6))
    A[17, 2] = np.inf
    S = sps.random(n, 9, density=0.2, format="csc", random_state=rng)
    X = tm.SplitMatrix([tm.DenseMatrix(A), tm.SparseMatrix(S), tm.CategoricalMatrix(rng.integers(0, 4, n))])
    cols = np.array([0, 1, 3, 4, 5, 6, 8, 15, 16])
    v = rng.standard_normal(X.shape[1])
    got = X.matvec(v, cols=cols)
    E = np.hstack([A, S.toarray(), np.eye(4)[X.matrices[2].indices]])
    want = E[:, cols] @ v[cols]
    assert np.isfinite(got).all() and np.abs(got - want).max() < 1e-12 * max(1, np.abs(want).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_densify_entry_points(dtype):
    """tm_csr_densify_cols_* / tm_csc_densify_cols_* / tm_dense_gather_cols_* against numpy."""
    import torch
    from tabmat_amd import _device as D
    from tabmat_amd.ext import dense as xd
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import CsrDev, DenseDev

    rng = np.random.default_rng(4)
    n, m = 3001, 77
    S = sps.random(n, m, density=0.07, format="csr", random_state=rng).astype(dtype)
    A = CsrDev.from_scipy(S)
    sel = np.sort(rng.choice(m, 9, replace=False)).astype(np.int32)
    want = S.toarray()[:, sel]
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    T = torch.zeros((n, 12), dtype=tdt, device="cuda")
    cmap = np.full(m, -1, dtype=np.int32)
    cmap[sel] = 2 + np.arange(9, dtype=np.int32)
    xs.csr_densify_cols(A, D.to_dev(cmap), T)
    got = T.cpu().numpy()
    assert np.array_equal(got[:, 2:11], want) and not got[:, :2].any() and not got[:, 11:].any()
    T.zero_()
    rws, vls, bstart, _, col_bptr = A.csc_blocks()
    cd = D.idx_dev(sel, torch.int64)
    seg = torch.stack([bstart[col_bptr[cd]], bstart[col_bptr[cd + 1]]], dim=1).contiguous()
    xs.csc_densify_cols(rws, vls, seg, D.to_dev(2 + np.arange(9, dtype=np.int32)),
                        int((seg[:, 1] - seg[:, 0]).max().item()), T)
    assert np.array_equal(T.cpu().numpy()[:, 2:11], want)
    X = rng.standard_normal((n, 20)).astype(dtype)
    pick = np.array([0, 3, 4, 19], dtype=np.int32)
    for arr in (X, np.asfortranarray(X)):
        T.zero_()
        xd.dense_gather_cols(DenseDev.from_host(arr), D.to_dev(pick), T, 5)
        got = T.cpu().numpy()
        assert np.array_equal(got[:, 5:9], X[:, pick]) and not got[:, :5].any() and not got[:, 9:].any()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_narrow_column_selection(dtype):
    """SparseMatrix.sandwich with a narrow `cols` (dense-block form from the CSC twin) vs the oracle,
    with and without rows, and equal to the generic path."""
    import tabmat_amd as tm
    import tabmat_amd.sparse_matrix as spm
    from oracle import oracle as orc

    rng = np.random.default_rng(9)
    n, m = 20_011, 700
    S = sps.random(n, m, density=0.03, format="csc", random_state=rng).astype(dtype)
    X = tm.SparseMatrix(S)
    X._narrow_pays = lambda w: True        # (the host cost model would send this small block the usual way)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 10)] = 0
    rows = np.sort(rng.choice(n, n // 4, replace=False))
    tol = 1e-10 if dtype == np.float64 else 3e-5
    assert not tm.SparseMatrix(S)._narrow_pays(23)     # 20k rows: the full product is cheaper
    for cols in (np.sort(rng.choice(m, 23, replace=False)), np.array([5]), np.arange(100, 228)):
        for r in (None, rows):
            got = X.sandwich(d, rows=r, cols=cols)
            want = orc.sparse_sandwich(S, S.tocsr(), d, r, cols)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    cols = np.sort(rng.choice(m, 40, replace=False))
    a = X.sandwich(d, cols=cols)
    old, spm.NARROW_COLS = spm.NARROW_COLS, 0
    try:
        b = X.sandwich(d, cols=cols)
    finally:
        spm.NARROW_COLS = old
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def test_column_selection_skips_unselected_heavy_blocks():
    """a selection that leaves out a high-cardinality categorical (a glum active set) must not
    compute -- or allocate the (p, p) result of -- the full product: the restricted path skips blocks without
    a selected column, like the reference (split_matrix.py:324-356)."""
    import tabmat_amd as tm
    import tabmat_amd.split_matrix as smod

    rng = np.random.default_rng(11)
    n = 6000
    X = rng.standard_normal((n, 140))
    S = sps.random(n, 60, density=0.1, format="csc", random_state=rng)
    big = rng.integers(0, 9000, n)
    small = rng.integers(0, 7, n)
    mat = tm.SplitMatrix([tm.DenseMatrix(X), tm.SparseMatrix(S),
                          tm.CategoricalMatrix(big, categories=np.arange(9000)),
                          tm.CategoricalMatrix(small, categories=np.arange(7))])
    p = mat.shape[1]
    # everything but the 9000-level categorical: > 128 dense + sparse columns and < half of all columns
    cols = np.concatenate([np.arange(200), np.arange(200 + 9000, p)]).astype(np.int32)
    assert not mat._full_product_pays(mat._split_col_subsets(cols)[1])
    seen = []
    orig = smod.D.zeros

    def spy(shape, dtype):
        seen.append(tuple(shape))
        return orig(shape, dtype)

    smod.D.zeros = spy
    try:
        d = rng.random(n)
        got = mat.sandwich(d, cols=cols)
    finally:
        smod.D.zeros = orig
    assert (p, p) not in seen, "the full (p, p) result was allocated"
    E = np.hstack([X, S.toarray(), np.eye(7)[small]])
    assert rel_err(got, E.T @ (d[:, None] * E)) < F64_TOL
    assert nat_err(got, E.T @ (d[:, None] * E)) < F64_TOL
    # a selection that covers every block still takes the tuned full product + selection
    cols2 = np.sort(rng.choice(p, size=int(0.7 * p), replace=False)).astype(np.int32)
    assert mat._full_product_pays(mat._split_col_subsets(cols2)[1]) == (p * p * 8 <= smod.FULL_RESULT_MAX_BYTES)
