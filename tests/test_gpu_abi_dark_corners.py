"""Entry points that the product's dispatch no longer (or only rarely) reaches -- found by the ABI call spy of round 6
(tests/test_zz_abi_coverage.py: 15 of 182 declared symbols were never called by 2536 GPU tests).  They stay in the
ABI (fallbacks for twins that are refused, float32 forms of the generic restricted kernels, the int32-column forms
next to the byte-column ones), so each is called here directly and compared with the oracle."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs  # noqa: F401
from _gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc

    return orc


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_generic_restricted_kernels_in_both_dtypes(dtype):
    """tm_sparse_sandwich_*, tm_csr_dense_sandwich_*, tm_cat_dense_sandwich_*, tm_cat_cat_sandwich_*,
    tm_cat_sparse_sandwich_*: the generic kernels with `rows` / `cols` lists (the reference's restricted loops:
    ext/sparse.pyx:17-77, ext/sparse_helpers-tmpl.cpp:23-146, ext/split.pyx:32-111, categorical_matrix.py:825-838).
    The float32 forms had no caller left in the suite: the tuned paths take float32 blocks elsewhere."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext import split as xsplit

    orc = _orc()
    rng = np.random.default_rng(41)
    n, m, k = 6_007, 90, 36
    S = sps.random(n, m, density=0.06, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.4
    S = S.astype(dtype)
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[::9] = 0
    ci = rng.integers(0, 13, n).astype(np.int32)
    cj = rng.integers(0, 7, n).astype(np.int32)
    ci[rng.integers(0, n, n // 12)] = -1                         # missing
    rows = np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
    Ac = np.sort(rng.choice(m, 40, replace=False)).astype(np.int32)
    Bc = np.sort(rng.choice(k, 17, replace=False)).astype(np.int32)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    A, Bd = sm._dev(), dm._dev()
    dd, rd, acd, bcd = _dev(d), D.idx_dev(rows), D.idx_dev(Ac), D.idx_dev(Bc)
    S64, B64, d64 = S.astype(np.float64), B.astype(np.float64), d.astype(np.float64)

    got = D.to_host(xs.sparse_sandwich(A, dd, rd, acd))
    assert rel_err(got, orc.sparse_sandwich(sps.csc_matrix(S64), sps.csr_matrix(S64), d64, rows, Ac)) < tol
    got = D.to_host(xs.csr_dense_sandwich(A, Bd, dd, rd, acd, bcd))
    assert rel_err(got, orc.csr_dense_sandwich(sps.csr_matrix(S64), B64, d64, rows, Ac, Bc)) < tol
    got = D.to_host(xsplit.sandwich_cat_dense(_dev(ci), 12, dd, Bd, rd, bcd, drop_first=True))
    assert rel_err(got, orc.sandwich_cat_dense(ci, 12, d64, B64, rows, Bc, drop_first=True)) < tol
    got = D.to_host(xsplit.sandwich_cat_cat(_dev(ci), _dev(cj), 13, 6, dd, rd, False, True))
    assert rel_err(got, orc.sandwich_cat_cat(ci, cj, 13, 6, d64, rows, False, True)) < tol
    got = D.to_host(xsplit.sandwich_cat_sparse(_dev(ci), 13, dd, A, rd, acd))
    assert rel_err(got, orc.sandwich_cat_sparse(ci, 13, d64, sps.csr_matrix(S64), rows, None, Ac)) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,r,dens", [(5000, 300, 256, 0.03), (777, 37, 100, 0.2), (4099, 513, 132, 0.01)])
def test_wide_interleaved_ell_kernel_called_directly(dtype, n, m, r, dens):
    """tm_csr_dense_sandwich_ellw_*: round 1's kernel on the wide interleaved-ELL twin, the fallback when both the
    entry twin and the lane-group twin are refused.  SparseMatrix._cross_sandwich has preferred the entry twin since
    round 4, so tests/test_gpu_kernels.py::test_csr_dense_sandwich_wide_ell no longer lands here."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import SlabEll

    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=dens, format="csc", random_state=rng).astype(dtype)
    B = rng.standard_normal((n, r)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[::7] = 0
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    tw = SlabEll.from_csr(sm._dev(), wide=True)
    assert tw is not None and tw.wide
    got = D.to_host(xs.csr_dense_sandwich_ell(tw, dm._dev_c(), _dev(d)))
    want = _orc().csr_dense_sandwich(S.tocsr().astype(np.float64), B.astype(np.float64), d.astype(np.float64),
                                     None, None, None)
    assert rel_err(got, want) < (1e-10 if dtype == np.float64 else 1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lane_group_kernel_with_column_sums_on_the_padded_stream(dtype):
    """tm_csr_dense_sandwich_lg_xtd_*: the lane-group kernel's A'd form on the PADDED stream (the twin is compacted by
    default since round 3, which routes every caller to tm_csr_dense_sandwich_lgc_*)."""
    import tabmat_amd as tm
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext._types import SlabLg

    rng = np.random.default_rng(9)
    n, m, k = 9_001, 70, 128
    dens = np.concatenate([np.full(60, 0.04), np.full(10, 0.3)])      # 10 columns overflow the fixed round
    Sd = np.where(rng.random((n, m)) < dens, rng.random((n, m)), 0.0)
    B = rng.standard_normal((n, k))
    d = rng.random(n)
    d[::7] = 0.0
    A = tm.SparseMatrix(sps.csc_matrix(Sd.astype(dtype)))
    Bd = tm.DenseMatrix(B.astype(dtype))
    lg = SlabLg.from_csr(A._dev(), max_pad=None)
    assert lg is not None and lg.cvals is None                        # not compacted
    out, csum = xs.csr_dense_sandwich_lg(lg, Bd._dev_c(), _dev(d.astype(dtype)), want_colsum=True)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    want = _orc().csr_dense_sandwich(sps.csr_matrix(Sd), B, d, None, None, None)
    assert rel_err(out.cpu().numpy(), want) < tol
    assert rel_err(csum.cpu().numpy(), Sd.T @ d) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_int32_column_forms_of_the_row_list_kernels(dtype):
    """tm_csr_dense_sandwich_rows_*, tm_multi_cat_sparse_sandwich_rows_*, tm_sparse_sandwich_chunked_rows_* with int32
    block columns: since round 6 the host keeps the chunk-major columns as bytes only and calls the _u8_ forms; the
    int32 forms (a C caller that holds a plain chunk-major CSR) must give the same numbers as the oracle."""
    import tabmat_amd as tm
    from tabmat_amd import _device as D
    from tabmat_amd._lib import call
    from tabmat_amd.ext import sparse as xs
    from tabmat_amd.ext import split as xsplit

    orc = _orc()
    rng = np.random.default_rng(17)
    n, m, k = 20_011, 300, 150
    S = sps.random(n, m, density=0.06, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    S = S.astype(dtype)
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 9)] = 0
    rows = rng.choice(n, n // 10, replace=False)
    rows = np.concatenate([rows, rows[:50]])                       # repeats: per occurrence outside the self term
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    A, Bd = sm._dev(), dm._dev_c()
    rows_d, dd = D.idx_dev(rows), _dev(d)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    suf = D.fsuf(A.data)
    c32 = A.chunk_cols32()
    S64, d64 = S.astype(np.float64), d.astype(np.float64)

    # sparse x dense over the row list
    cm_data, _, ranges, r32, d_sel = xs._row_table(A, rows_d, dd, False)
    out = D.zeros((A.m, Bd.m), A.dtype)
    call(f"tm_csr_dense_sandwich_rows_{suf}", D.p(cm_data), D.p(c32), D.p(ranges), int(r32.numel()), D.p(r32),
         D.p(d_sel), A.n, A.m, D.p(Bd.buf), Bd.m, Bd.order_f, D.p(out), D.stream_ptr())
    want = orc.csr_dense_sandwich(sps.csr_matrix(S64), B.astype(np.float64), d64, rows, None, None)
    assert rel_err(D.to_host(out), want) < tol
    assert rel_err(D.to_host(xs.csr_dense_sandwich_rows(A, Bd, dd, rows_d)), want) < tol          # (_u8_ form)

    # categorical x sparse over the row list
    levels, drops = (13, 40, 5), (False, True, False)
    codes = [rng.integers(0, L, n).astype(np.int32) for L in levels]
    cats = [(_dev(c), L - int(dr), dr) for c, L, dr in zip(codes, levels, drops)]
    total = sum(c[1] for c in cats)
    res = D.out_buf((total, A.m), A.data.dtype)
    cargs = xsplit._cat_args(cats)
    call(f"tm_multi_cat_sparse_sandwich_rows_{suf}", *cargs, D.p(cm_data), D.p(c32), D.p(ranges), D.p(r32),
         int(r32.numel()), A.m, D.p(d_sel), D.p(res), D.stream_ptr())
    want = np.vstack([orc.sandwich_cat_sparse(c, L - int(dr), d64, sps.csr_matrix(S64), rows.astype(np.int32), None,
                                              None, dr) for c, L, dr in zip(codes, levels, drops)])
    assert rel_err(D.to_host(res), want) < tol

    # sparse self over the row list (a row SET: ext/sparse.pyx:46-48), int32 columns through the A/B switch
    want = orc.sparse_sandwich(sps.csc_matrix(S64), sps.csr_matrix(S64), d64, rows, None)
    old = xs.K2B_U8
    xs.K2B_U8 = False
    try:
        got32 = D.to_host(xs.sparse_sandwich_rows(A, dd, rows_d))
    finally:
        xs.K2B_U8 = old
    assert rel_err(got32, want) < tol
    assert rel_err(D.to_host(xs.sparse_sandwich_rows(A, dd, rows_d)), want) < tol
