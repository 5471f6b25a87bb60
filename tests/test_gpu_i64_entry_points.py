"""The int64-column-index forms of the sparse entry points (tm_*_i64_*, csrc/sparse_i64.hip; reference:
ext/sparse.pyx:13-15 `win_integral` -- int32 or int64 index arrays): called straight through the C ABI with int64
device arrays, against the int32 symbols and dense algebra; an out-of-range index is clamped and reported."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy import sparse as sps

from _gpu_util import nat_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_i64_forms_match_dense_algebra(dtype):
    from tabmat_amd import _device as D
    from tabmat_amd._lib import call

    rng = np.random.default_rng(3)
    n, m, r = 7000, 300, 40
    S = sps.random(n, m, density=0.04, format="csr", random_state=rng).astype(dtype)
    S.sort_indices()
    suf = "f64" if dtype == np.float64 else "f32"
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    data = torch.from_numpy(S.data.copy()).cuda()
    ind64 = torch.from_numpy(S.indices.astype(np.int64)).cuda()
    ptr = torch.from_numpy(S.indptr.astype(np.int64)).cuda()
    nnz = int(S.nnz)
    d = torch.from_numpy(rng.random(n).astype(dtype)).cuda()
    v = torch.from_numpy(rng.standard_normal(m).astype(dtype)).cuda()
    B = torch.from_numpy(rng.standard_normal((n, r)).astype(dtype)).cuda()
    st = D.stream_ptr()
    S64 = S.astype(np.float64)
    dh, vh, Bh = d.cpu().numpy().astype(np.float64), v.cpu().numpy().astype(np.float64), B.cpu().numpy().astype(np.float64)
    tol = 1e-10 if dtype == np.float64 else 3e-5

    out = torch.empty((m, m), dtype=tdt, device="cuda")
    call(f"tm_sparse_sandwich_i64_{suf}", D.p(data), D.p(ind64), D.p(ptr), n, m, nnz, D.p(d), None, 0, None, 0,
         D.p(out), st)
    assert nat_err(out.cpu().numpy(), (S64.T.multiply(dh)).dot(S64).toarray()) < tol

    out = torch.empty((m, r), dtype=tdt, device="cuda")
    call(f"tm_csr_dense_sandwich_i64_{suf}", D.p(data), D.p(ind64), D.p(ptr), n, m, nnz, D.p(B), r, 0, D.p(d),
         None, 0, None, 0, None, 0, D.p(out), st)
    want = S64.T.dot(dh[:, None] * Bh)
    assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < tol

    out = torch.zeros((n,), dtype=tdt, device="cuda")
    call(f"tm_csr_matvec_i64_{suf}", D.p(data), D.p(ind64), D.p(ptr), n, m, nnz, D.p(v), None, 0, None, 0,
         D.p(out), st)
    want = S64.dot(vh)
    assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < tol

    out = torch.zeros((m,), dtype=tdt, device="cuda")
    call(f"tm_csr_rmatvec_i64_{suf}", D.p(data), D.p(ind64), D.p(ptr), n, m, nnz, D.p(d), None, 0, None, 0,
         D.p(out), st)
    want = S64.T.dot(dh)
    assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < tol

    out = torch.zeros((m,), dtype=tdt, device="cuda")
    call(f"tm_csr_col_sq_i64_{suf}", D.p(data), D.p(ind64), D.p(ptr), n, m, nnz, D.p(d), D.p(out), st)
    want = S64.multiply(S64).T.dot(dh)
    assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < tol

    bad = C.c_int32(7)
    call("tm_index_check_i64", st, C.byref(bad))
    assert bad.value == 0

    # an index beyond the column count: clamped (no out-of-bounds access), reported once, flag cleared
    ind_bad = ind64.clone()
    ind_bad[5] = m + 12345678901
    out = torch.zeros((m,), dtype=tdt, device="cuda")
    call(f"tm_csr_rmatvec_i64_{suf}", D.p(data), D.p(ind_bad), D.p(ptr), n, m, nnz, D.p(d), None, 0, None, 0,
         D.p(out), st)
    call("tm_index_check_i64", st, C.byref(bad))
    assert bad.value == 1
    call("tm_index_check_i64", st, C.byref(bad))
    assert bad.value == 0
