"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle.

Loads ``oracle/_build/liboracle.so`` (built by ``make -C oracle``; C restatement
of the reference's native loops, see ``oracle_kernels.inc.h``) and restates the
thin Python/Cython layers above them (argument normalisation, block assembly)
in numpy.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module -- as the checker or the timed CPU
baseline, never as a product path.  ``tabmat_amd`` must not import it.

Parity status: pinned against the reference's own known-answer tests (restated
in ``tests/test_oracle_*.py``) and its data fixture ``tests/real_matrix.pkl``
(committed as ``tests/golden/real_matrix_blocks.npz``).  No ``oracle/_ref``
build exists: the reference's native code needs mako, xsimd and jemalloc, which
this image lacks.

All file:line citations are relative to ``/root/reference/``.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np
from scipy import sparse as sps

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when up to date)."""
    srcs = [os.path.join(_HERE, f) for f in ("tabmat_oracle.c", "oracle_kernels.inc.h")]
    if (
        force
        or not os.path.exists(_SO)
        or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
_FS = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}
_IS = {np.dtype(np.int32): "i32", np.dtype(np.int64): "i64"}


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _i64(x) -> C.c_int64:
    return C.c_int64(int(x))


def _fs(dtype) -> str:
    try:
        return _FS[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"oracle supports float32/float64 only, got {dtype}")


def _order_f(X: np.ndarray) -> int:
    if X.flags["C_CONTIGUOUS"]:
        return 0
    if X.flags["F_CONTIGUOUS"]:
        return 1
    raise Exception("The matrix X is not contiguous.")  # ext/dense.pyx:43


def set_up_rows_or_cols(arr, length: int, dtype=np.int32) -> np.ndarray:
    """util.py:6-12."""
    if arr is None:
        return np.arange(length, dtype=dtype)
    return np.ascontiguousarray(np.asarray(arr).astype(dtype))


def _call(name: str, *args):
    fn = getattr(lib(), name)
    fn.restype = None
    fn(*args)


# --------------------------------------------------------------------------- #
# ext.dense  (ext/dense.pyx:19-101)
# --------------------------------------------------------------------------- #
def dense_sandwich(X: np.ndarray, d: np.ndarray, rows, cols) -> np.ndarray:
    n, m = X.shape
    rows = set_up_rows_or_cols(rows, n)
    cols = set_up_rows_or_cols(cols, m)
    out = np.zeros((len(cols), len(cols)), dtype=X.dtype)
    if len(rows) == 0 or len(cols) == 0:
        return out
    d = np.ascontiguousarray(d, dtype=X.dtype)
    _call(f"orc_dense_sandwich_{_fs(X.dtype)}", _p(X), _i64(n), _i64(m), C.c_int(_order_f(X)),
          _p(d), _p(rows), _i64(len(rows)), _p(cols), _i64(len(cols)), _p(out))
    return out


def _dense_mv(kind: str, X, v, rows, cols) -> np.ndarray:
    n, m = X.shape
    rows = set_up_rows_or_cols(rows, n)
    cols = set_up_rows_or_cols(cols, m)
    out = np.zeros(len(cols) if kind == "rmatvec" else len(rows), dtype=X.dtype)
    if len(rows) == 0 or len(cols) == 0:
        return out
    v = np.ascontiguousarray(v, dtype=X.dtype)
    _call(f"orc_dense_{kind}_{_fs(X.dtype)}", _p(X), _i64(n), _i64(m), C.c_int(_order_f(X)),
          _p(v), _p(rows), _i64(len(rows)), _p(cols), _i64(len(cols)), _p(out))
    return out


def dense_rmatvec(X, v, rows, cols) -> np.ndarray:
    return _dense_mv("rmatvec", X, v, rows, cols)


def dense_matvec(X, v, rows, cols) -> np.ndarray:
    return _dense_mv("matvec", X, v, rows, cols)


def dense_col_sq_dev(X, w, shift) -> np.ndarray:
    """ext/dense.pyx:103-122 (transpose_square_dot_weights)."""
    out = np.zeros(X.shape[1], dtype=X.dtype)
    _call(f"orc_dense_col_sq_dev_{_fs(X.dtype)}", _p(X), _i64(X.shape[0]), _i64(X.shape[1]),
          C.c_int(_order_f(X)), _p(np.ascontiguousarray(w, dtype=X.dtype)),
          _p(np.ascontiguousarray(shift, dtype=X.dtype)), _p(out))
    return out


# --------------------------------------------------------------------------- #
# ext.sparse  (ext/sparse.pyx)
# --------------------------------------------------------------------------- #
def _sp_parts(A):
    data = np.ascontiguousarray(A.data)
    idt = np.dtype(max(A.indices.dtype, A.indptr.dtype))
    return data, np.ascontiguousarray(A.indices, dtype=idt), np.ascontiguousarray(A.indptr, dtype=idt)


def _fi(data, ind) -> str:
    return f"{_fs(data.dtype)}_{_IS[np.dtype(ind.dtype)]}"


def sparse_sandwich(A_csc, AT_csr, d, rows, cols) -> np.ndarray:
    """ext/sparse.pyx:17-77."""
    Ad, Ai, Ap = _sp_parts(A_csc)
    Td, Ti, Tp = _sp_parts(AT_csr)
    Ti = Ti.astype(Ai.dtype, copy=False)
    Tp = Tp.astype(Ai.dtype, copy=False)
    n, ncol = A_csc.shape
    rows = set_up_rows_or_cols(rows, n, Ai.dtype)
    cols = set_up_rows_or_cols(cols, ncol, Ai.dtype)
    m = len(cols)
    out = np.zeros((m, m), dtype=A_csc.dtype)
    d = np.ascontiguousarray(d, dtype=A_csc.dtype)
    _call(f"orc_sparse_sandwich_{_fi(Ad, Ai)}", _p(Ad), _p(Ai), _p(Ap), _p(Td), _p(Ti), _p(Tp),
          _i64(n), _i64(ncol), _p(d), _p(rows), _i64(len(rows)), _p(cols), _i64(m), _p(out))
    return out


def csr_dense_sandwich(A_csr, B: np.ndarray, d, rows, A_cols, B_cols) -> np.ndarray:
    """ext/sparse.pyx:211-260."""
    Ad, Ai, Ap = _sp_parts(A_csr)
    n, m = A_csr.shape
    r = B.shape[1]
    rows = set_up_rows_or_cols(rows, n, Ai.dtype)
    A_cols = set_up_rows_or_cols(A_cols, m, Ai.dtype)
    B_cols = set_up_rows_or_cols(B_cols, r, Ai.dtype)
    out = np.zeros((len(A_cols), len(B_cols)), dtype=A_csr.dtype)
    if len(rows) == 0 or len(A_cols) == 0 or len(B_cols) == 0 or A_csr.nnz == 0:
        return out
    d = np.ascontiguousarray(d, dtype=A_csr.dtype)
    _call(f"orc_csr_dense_sandwich_{_fi(Ad, Ai)}", _p(Ad), _p(Ai), _p(Ap), _p(B), _p(d), _p(out),
          _i64(m), _i64(n), _i64(r), C.c_int(_order_f(B)), _p(rows), _p(A_cols), _p(B_cols),
          _i64(len(rows)), _i64(len(A_cols)), _i64(len(B_cols)))
    return out


def csr_matvec_unrestricted(X_csr, v, out=None) -> np.ndarray:
    Xd, Xi, Xp = _sp_parts(X_csr)
    if out is None:
        out = np.zeros(X_csr.shape[0], dtype=X_csr.dtype)
    v = np.ascontiguousarray(v, dtype=X_csr.dtype)
    _call(f"orc_csr_matvec_unrestricted_{_fi(Xd, Xi)}", _p(Xd), _p(Xi), _p(Xp),
          _i64(X_csr.shape[0]), _p(v), _p(out))
    return out


def csr_matvec(X_csr, v, rows, cols) -> np.ndarray:
    Xd, Xi, Xp = _sp_parts(X_csr)
    rows = set_up_rows_or_cols(rows, X_csr.shape[0], Xi.dtype)
    cols = set_up_rows_or_cols(cols, X_csr.shape[1], Xi.dtype)
    out = np.zeros(len(rows), dtype=X_csr.dtype)
    v = np.ascontiguousarray(v, dtype=X_csr.dtype)
    _call(f"orc_csr_matvec_{_fi(Xd, Xi)}", _p(Xd), _p(Xi), _p(Xp), _i64(X_csr.shape[1]), _p(v),
          _p(rows), _i64(len(rows)), _p(cols), _i64(len(cols)), _p(out))
    return out


def csc_rmatvec_unrestricted(XT_csc, v, out=None) -> np.ndarray:
    Xd, Xi, Xp = _sp_parts(XT_csc)
    if out is None:
        out = np.zeros(XT_csc.shape[1], dtype=XT_csc.dtype)
    v = np.ascontiguousarray(v, dtype=XT_csc.dtype)
    _call(f"orc_csc_rmatvec_unrestricted_{_fi(Xd, Xi)}", _p(Xd), _p(Xi), _p(Xp),
          _i64(XT_csc.shape[1]), _p(v), _p(out))
    return out


def csc_rmatvec(XT_csc, v, rows, cols) -> np.ndarray:
    Xd, Xi, Xp = _sp_parts(XT_csc)
    rows = set_up_rows_or_cols(rows, XT_csc.shape[0], Xi.dtype)
    cols = set_up_rows_or_cols(cols, XT_csc.shape[1], Xi.dtype)
    out = np.zeros(len(cols), dtype=XT_csc.dtype)
    v = np.ascontiguousarray(v, dtype=XT_csc.dtype)
    _call(f"orc_csc_rmatvec_{_fi(Xd, Xi)}", _p(Xd), _p(Xi), _p(Xp), _i64(XT_csc.shape[0]), _p(v),
          _p(rows), _i64(len(rows)), _p(cols), _i64(len(cols)), _p(out))
    return out


def csc_col_sq(A_csc, w) -> np.ndarray:
    """ext/sparse.pyx:262-282."""
    Ad, Ai, Ap = _sp_parts(A_csc)
    out = np.zeros(A_csc.shape[1], dtype=A_csc.dtype)
    _call(f"orc_csc_col_sq_{_fi(Ad, Ai)}", _p(Ad), _p(Ai), _p(Ap), _i64(A_csc.shape[1]),
          _p(np.ascontiguousarray(w, dtype=A_csc.dtype)), _p(out))
    return out


# --------------------------------------------------------------------------- #
# ext.categorical / ext.split
# --------------------------------------------------------------------------- #
def _col_included(cols, n_cols) -> Optional[np.ndarray]:
    """ext/categorical.pyx:120-125 (get_col_included)."""
    if cols is None:
        return None
    inc = np.zeros(max(n_cols, 1), dtype=np.int32)
    inc[np.asarray(cols, dtype=np.int64)] = 1
    return inc


def cat_transpose_matvec(indices, other, n_cols, rows, cols, out, drop_first=False) -> None:
    """ext/categorical.pyx:23-117 (transpose_matvec_fast/_complex): in-place +=."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indices)
    if rows is not None and len(rows) == n:
        rows = None
    if cols is not None and len(cols) == n_cols:
        cols = None
    rows_a = None if rows is None else set_up_rows_or_cols(rows, n)
    inc = _col_included(cols, n_cols)
    other = np.ascontiguousarray(other, dtype=out.dtype)
    _call(f"orc_cat_transpose_matvec_{_fs(out.dtype)}", _p(indices), _i64(n), _p(other),
          _i64(n_cols), C.c_int(int(drop_first)), _p(rows_a),
          _i64(0 if rows_a is None else len(rows_a)), _p(inc), _p(out))


def cat_matvec(indices, other, n_rows, cols, n_cols, out_vec, drop_first=False) -> None:
    """ext/categorical.pyx:128-180 (matvec_fast/_complex): in-place +=."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    inc = _col_included(cols, n_cols)
    other = np.ascontiguousarray(other, dtype=out_vec.dtype)
    _call(f"orc_cat_matvec_{_fs(out_vec.dtype)}", _p(indices), _i64(n_rows), _p(other),
          C.c_int(int(drop_first)), _p(inc), _p(out_vec))


def sandwich_categorical(indices, d, rows, n_cols, drop_first=False) -> np.ndarray:
    """ext/categorical.pyx:183-218."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    rows = set_up_rows_or_cols(rows, len(indices))
    d = np.ascontiguousarray(d)
    res = np.zeros(n_cols, dtype=d.dtype)
    _call(f"orc_cat_sandwich_diag_{_fs(d.dtype)}", _p(indices), _p(d), _p(rows), _i64(len(rows)),
          C.c_int(int(drop_first)), _p(res))
    return res


def sandwich_cat_cat(i_indices, j_indices, i_ncol, j_ncol, d, rows,
                     i_drop_first=False, j_drop_first=False) -> np.ndarray:
    """ext/split.pyx:83-111."""
    i_indices = np.ascontiguousarray(i_indices, dtype=np.int32)
    j_indices = np.ascontiguousarray(j_indices, dtype=np.int32)
    rows = set_up_rows_or_cols(rows, len(i_indices))
    d = np.ascontiguousarray(d)
    res = np.zeros((i_ncol, j_ncol), dtype=d.dtype)
    _call(f"orc_cat_cat_sandwich_{_fs(d.dtype)}", _p(i_indices), _p(j_indices), _p(d), _p(rows),
          _i64(len(rows)), _i64(i_ncol), _i64(j_ncol), C.c_int(int(i_drop_first)),
          C.c_int(int(j_drop_first)), _p(res))
    return res


def sandwich_cat_dense(i_indices, i_ncol, d, mat_j: np.ndarray, rows, j_cols,
                       drop_first=False) -> np.ndarray:
    """ext/split.pyx:32-80."""
    i_indices = np.ascontiguousarray(i_indices, dtype=np.int32)
    rows = set_up_rows_or_cols(rows, len(i_indices))
    j_cols = set_up_rows_or_cols(j_cols, mat_j.shape[1])
    res = np.zeros((i_ncol, len(j_cols)), dtype=mat_j.dtype)
    if len(d) == 0 or len(rows) == 0 or len(j_cols) == 0 or i_ncol == 0:
        return res
    d = np.ascontiguousarray(d, dtype=mat_j.dtype)
    _call(f"orc_cat_dense_sandwich_{_fs(mat_j.dtype)}", _p(i_indices), _p(d), _p(rows),
          _i64(len(rows)), _p(j_cols), _i64(len(j_cols)), _p(mat_j), _i64(mat_j.shape[0]),
          _i64(mat_j.shape[1]), C.c_int(_order_f(mat_j)), C.c_int(int(drop_first)), _i64(i_ncol),
          _p(res))
    return res


def sandwich_cat_sparse(i_indices, i_ncol, d, S_csr, rows, L_cols, R_cols,
                        drop_first=False) -> np.ndarray:
    """categorical_matrix.py:825-838 (_cross_sparse; scipy.sparse matmul)."""
    i_indices = np.ascontiguousarray(i_indices, dtype=np.int32)
    Sd, Si, Sp = _sp_parts(S_csr)
    rows = set_up_rows_or_cols(rows, len(i_indices), Si.dtype)
    R = set_up_rows_or_cols(R_cols, S_csr.shape[1], Si.dtype)
    res = np.zeros((i_ncol, len(R)), dtype=S_csr.dtype)
    d = np.ascontiguousarray(d, dtype=S_csr.dtype)
    _call(f"orc_cat_sparse_sandwich_{_fi(Sd, Si)}", _p(i_indices), C.c_int(int(drop_first)),
          _i64(i_ncol), _p(Sd), _p(Si), _p(Sp), _i64(S_csr.shape[1]), _p(d), _p(rows),
          _i64(len(rows)), _p(R), _i64(len(R)), _p(res))
    if L_cols is not None and len(L_cols) < i_ncol:
        res = res[np.asarray(L_cols, dtype=np.int64)]
    return res


def split_col_subsets(indices: Sequence[np.ndarray], cols) -> tuple[list, list, int]:
    """ext/split.pyx:157-209."""
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    nb = len(indices)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(i, dtype=np.int64) for i in indices])
                                if nb else np.zeros(0, np.int64))
    offs = np.zeros(nb + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(i) for i in indices])
    sub_idx = np.zeros(max(len(cols), 1), dtype=np.int32)
    sub_cols = np.zeros(max(len(cols), 1), dtype=np.int32)
    counts = np.zeros(nb, dtype=np.int64)
    starts = np.zeros(nb, dtype=np.int64)
    _call("orc_split_col_subsets", _p(flat), _p(offs), _i64(nb), _p(cols), _i64(len(cols)),
          _p(sub_idx), _p(sub_cols), _p(counts), _p(starts))
    a = [sub_idx[starts[b]:starts[b] + counts[b]].copy() for b in range(nb)]
    b_ = [sub_cols[starts[b]:starts[b] + counts[b]].copy() for b in range(nb)]
    return a, b_, len(cols)


# --------------------------------------------------------------------------- #
# Block descriptors + SplitMatrix-level restatement (split_matrix.py:324-460)
# --------------------------------------------------------------------------- #
class Dense:
    kind = "dense"

    def __init__(self, X):
        X = np.asarray(X)
        if not (X.flags["C_CONTIGUOUS"] or X.flags["F_CONTIGUOUS"]):
            X = np.asfortranarray(X)  # dense_matrix.py:47-58
        self.X = X
        self.shape = X.shape
        self.dtype = X.dtype

    def toarray(self):
        return self.X


class Sparse:
    kind = "sparse"

    def __init__(self, A):
        A = sps.csc_matrix(A)
        if not A.has_sorted_indices:
            A.sort_indices()
        self.csc = A
        self.csr = A.tocsr()
        self.shape = A.shape
        self.dtype = A.dtype

    def toarray(self):
        return self.csc.toarray()


class Cat:
    kind = "cat"

    def __init__(self, codes, n_categories, drop_first=False, dtype=np.float64):
        self.codes = np.ascontiguousarray(codes, dtype=np.int32)
        self.n_categories = int(n_categories)
        self.drop_first = bool(drop_first)
        self.shape = (len(self.codes), max(self.n_categories - int(drop_first), 0))
        self.dtype = np.dtype(dtype)

    def toarray(self):
        out = np.zeros(self.shape, dtype=self.dtype)
        c = self.codes.astype(np.int64) - int(self.drop_first)
        ok = c >= 0
        out[np.nonzero(ok)[0], c[ok]] = 1
        return out


def block_sandwich(b, d, rows, cols):
    """Diagonal block: returns dense (k,k) or, for Cat, the diagonal vector."""
    if b.kind == "dense":
        return dense_sandwich(b.X, d, rows, cols)
    if b.kind == "sparse":
        return sparse_sandwich(b.csc, b.csr, d, rows, cols)
    diag = sandwich_categorical(b.codes, d, rows, b.shape[1], b.drop_first)
    if cols is not None and len(cols) < b.shape[1]:
        diag = diag[np.asarray(cols, dtype=np.int64)]
    return diag


def _rc(res, L_cols, R_cols):
    """categorical_matrix.py:296-316 (_row_col_indexing)."""
    if L_cols is not None and len(L_cols) != res.shape[0]:
        res = res[np.asarray(L_cols, dtype=np.int64)]
    if R_cols is not None and len(R_cols) != res.shape[1]:
        res = res[:, np.asarray(R_cols, dtype=np.int64)]
    return res


def cross_sandwich(bi, bj, d, rows, L_cols, R_cols):
    """<block>._cross_sandwich dispatch (dense_matrix.py:165-178,
    sparse_matrix.py:187-204, categorical_matrix.py:655-671)."""
    ki, kj = bi.kind, bj.kind
    if ki == "dense":
        if kj in ("sparse", "cat"):
            return cross_sandwich(bj, bi, d, rows, R_cols, L_cols).T
        raise TypeError
    if ki == "sparse":
        if kj == "dense":
            return csr_dense_sandwich(bi.csr, bj.X, d, rows, L_cols, R_cols)
        if kj == "cat":
            return cross_sandwich(bj, bi, d, rows, R_cols, L_cols).T
        raise TypeError
    # ki == cat
    if kj == "dense":
        res = sandwich_cat_dense(bi.codes, bi.shape[1], d, bj.X, rows, R_cols, bi.drop_first)
        return _rc(res, L_cols, None)
    if kj == "sparse":
        return sandwich_cat_sparse(bi.codes, bi.shape[1], d, bj.csr, rows, L_cols, R_cols,
                                   bi.drop_first)
    res = sandwich_cat_cat(bi.codes, bj.codes, bi.shape[1], bj.shape[1], d, rows,
                           bi.drop_first, bj.drop_first)
    return _rc(res, L_cols, R_cols)


def split_sandwich(blocks, indices, d, rows=None, cols=None) -> np.ndarray:
    """split_matrix.py:324-356: float64 (n_cols, n_cols) assembled from block results."""
    d = np.asarray(d)
    if cols is None:
        sub_idx = [np.asarray(i, dtype=np.int64) for i in indices]
        sub_cols = [None] * len(indices)
        n_cols = sum(len(i) for i in indices)
    else:
        sub_idx, sub_cols, n_cols = split_col_subsets(indices, set_up_rows_or_cols(cols, 0))
    out = np.zeros((n_cols, n_cols))
    for i, bi in enumerate(blocks):
        idx_i = sub_idx[i]
        res = block_sandwich(bi, d, rows, sub_cols[i])
        if bi.kind == "cat":
            out[(idx_i, idx_i)] += res
        else:
            out[np.ix_(idx_i, idx_i)] = res
        for j in range(i + 1, len(blocks)):
            idx_j = sub_idx[j]
            res = cross_sandwich(bi, blocks[j], d, rows, sub_cols[i], sub_cols[j])
            out[np.ix_(idx_i, idx_j)] = res
            out[np.ix_(idx_j, idx_i)] = res.T
    return out


def block_matvec(b, v, cols, out):
    """out += b[:, cols] @ v[cols] with v of full block width."""
    n = b.shape[0]
    if b.kind == "dense":
        if cols is None or len(cols) == b.shape[1]:
            out += b.X.dot(v)  # dense_matrix.py:212-217 (BLAS gemv)
        else:
            out += dense_matvec(b.X, v, None, cols)
    elif b.kind == "sparse":
        if cols is None or len(cols) == b.shape[1]:
            csr_matvec_unrestricted(b.csr, v, out)
        else:
            out += csr_matvec(b.csr, v, None, cols)
    else:
        c = None if (cols is None or len(cols) == b.shape[1]) else cols
        cat_matvec(b.codes, v, n, c, b.shape[1], out, b.drop_first)
    return out


def block_transpose_matvec(b, v, rows, cols):
    """returns b[rows, cols].T @ v[rows] (length len(cols))."""
    n, m = b.shape
    unr = (rows is None or len(rows) == n) and (cols is None or len(cols) == m)
    if b.kind == "dense":
        if unr:
            return b.X.T.dot(v)
        return dense_rmatvec(b.X, v, rows, cols)
    if b.kind == "sparse":
        if unr:
            return csc_rmatvec_unrestricted(b.csc, v)
        return csc_rmatvec(b.csc, v, rows, cols)
    out = np.zeros(m, dtype=np.asarray(v).dtype)
    cat_transpose_matvec(b.codes, v, m, rows, cols, out, b.drop_first)
    if cols is not None:
        return out[np.asarray(cols, dtype=np.int64)]
    return out


def split_matvec(blocks, indices, v, cols=None) -> np.ndarray:
    """split_matrix.py:373-420."""
    v = np.asarray(v)
    n = blocks[0].shape[0]
    if cols is None:
        sub_cols = [None] * len(indices)
    else:
        _, sub_cols, _ = split_col_subsets(indices, set_up_rows_or_cols(cols, 0))
    out = np.zeros(n, dtype=np.result_type(blocks[0].dtype, v.dtype))
    for b, idx, sc in zip(blocks, indices, sub_cols):
        block_matvec(b, np.ascontiguousarray(v[np.asarray(idx)], dtype=out.dtype), sc, out)
    return out


def split_transpose_matvec(blocks, indices, v, rows=None, cols=None) -> np.ndarray:
    """split_matrix.py:422-460 (out=None form)."""
    v = np.asarray(v)
    if cols is None:
        sub_idx = [np.asarray(i, dtype=np.int64) for i in indices]
        sub_cols = [None] * len(indices)
        n_cols = sum(len(i) for i in indices)
    else:
        sub_idx, sub_cols, n_cols = split_col_subsets(indices, set_up_rows_or_cols(cols, 0))
    out = np.zeros(n_cols, dtype=np.result_type(blocks[0].dtype, v.dtype))
    vv = np.ascontiguousarray(v, dtype=out.dtype)
    for b, idx, sc in zip(blocks, sub_idx, sub_cols):
        out[idx] += block_transpose_matvec(b, vv, rows, sc)
    return out


def split_toarray(blocks, indices) -> np.ndarray:
    n = blocks[0].shape[0]
    p = sum(len(i) for i in indices)
    out = np.empty((n, p))
    for b, idx in zip(blocks, indices):
        out[:, np.asarray(idx)] = b.toarray()
    return out
