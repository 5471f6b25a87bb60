"""Mirror of tabmat.ext.dense (reference: src/tabmat/ext/dense.pyx)."""
from __future__ import annotations

from .. import _device as D
from .._lib import call
from ._types import DenseDev


def dense_sandwich(X: DenseDev, d, rows, cols, center=None):
    """ext/dense.pyx:19-44.  rows/cols: int32 device tensors or None (= all).
    center (device tensor of the block's dtype, length X.m, or None): the product of X - 1 center'
    (tm_dense_sandwich_centered_*: the columns are centred on the way in)."""
    out_m = X.m if cols is None else D.nlen(cols)
    in_n = X.n if rows is None else D.nlen(rows)
    if in_n == 0 or out_m == 0:  # ext/dense.pyx:26-27
        return D.zeros((out_m, out_m), X.dtype)
    out = D.out_buf((out_m, out_m), X.dtype)
    D.same_float("dense_sandwich", X.buf, d)
    if center is not None:
        D.same_float("dense_sandwich", X.buf, center)
        assert center.numel() == X.m and center.is_contiguous()
        call(f"tm_dense_sandwich_centered_{D.fsuf(X.buf)}", D.p(X.buf), X.n, X.m, X.order_f, D.p(d), D.p(rows),
             D.nlen(rows), D.p(cols), D.nlen(cols), D.p(center), D.p(out), D.stream_ptr())
        return out
    call(f"tm_dense_sandwich_{D.fsuf(X.buf)}", D.p(X.buf), X.n, X.m, X.order_f, D.p(d), D.p(rows),
         D.nlen(rows), D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def dense_rmatvec(X: DenseDev, v, rows, cols, out=None):
    """ext/dense.pyx:48-73: X[rows, cols].T @ v[rows] (length len(cols)); accumulates into out."""
    n_cols = X.m if cols is None else D.nlen(cols)
    n_rows = X.n if rows is None else D.nlen(rows)
    if out is None:
        out = D.zeros((n_cols,), X.dtype)
    if n_rows == 0 or n_cols == 0:
        return out
    D.same_float("dense_rmatvec", X.buf, v, out)
    call(f"tm_dense_rmatvec_{D.fsuf(X.buf)}", D.p(X.buf), X.n, X.m, X.order_f, D.p(v), D.p(rows),
         D.nlen(rows), D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def dense_matvec(X: DenseDev, v, rows, cols, out=None):
    """ext/dense.pyx:76-101: X[rows, cols] @ v[cols] (length len(rows)); accumulates into out."""
    n_cols = X.m if cols is None else D.nlen(cols)
    n_rows = X.n if rows is None else D.nlen(rows)
    if out is None:
        out = D.zeros((n_rows,), X.dtype)
    if n_rows == 0 or n_cols == 0:
        return out
    D.same_float("dense_matvec", X.buf, v, out)
    call(f"tm_dense_matvec_{D.fsuf(X.buf)}", D.p(X.buf), X.n, X.m, X.order_f, D.p(v), D.p(rows),
         D.nlen(rows), D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def dense_matvec_multi(X: DenseDev, V, rows, cols, transpose):
    """2-D operand: X[rows, cols] @ V[cols] (n_rows, K) or X[rows, cols].T @ V[rows] (n_cols, K);
    the reference leaves these to NumPy BLAS (dense_matrix.py:212-217)."""
    K = int(V.shape[1])
    n_cols = X.m if cols is None else D.nlen(cols)
    n_rows = X.n if rows is None else D.nlen(rows)
    out = D.zeros((n_cols if transpose else n_rows, K), X.dtype)
    if K == 0 or n_rows == 0 or n_cols == 0:
        return out
    D.same_float("dense_matvec_multi", X.buf, V)
    Vc = V.contiguous()
    fn = "tm_dense_rmatvec_multi_" if transpose else "tm_dense_matvec_multi_"
    call(fn + D.fsuf(X.buf), D.p(X.buf), X.n, X.m, X.order_f, D.p(Vc), K, D.p(rows), D.nlen(rows),
         D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def transpose_square_dot_weights(X: DenseDev, weights, shift):
    """ext/dense.pyx:103-122: out[j] = sum_i w[i] * (X[i, j] - shift[j])**2."""
    out = D.zeros((X.m,), X.dtype)
    if X.n == 0 or X.m == 0:
        return out
    D.same_float("transpose_square_dot_weights", X.buf, weights, shift)
    call(f"tm_dense_col_sq_dev_{D.fsuf(X.buf)}", D.p(X.buf), X.n, X.m, X.order_f, D.p(weights),
         D.p(shift), D.p(out), D.stream_ptr())
    return out


def dense_gather_cols(X: DenseDev, cols, T, t0: int):
    """T[:, t0 + q] = X[:, cols[q]] (cols: int32 device tensor); T: (n, ld) row-major device tensor."""
    D.same_float("dense_gather_cols", X.buf, T)
    call(f"tm_dense_gather_cols_{D.fsuf(T)}", D.p(X.buf), X.n, X.m, int(X.order_f), D.p(cols),
         D.nlen(cols), D.p(T), T.shape[1], int(t0), D.stream_ptr())


def dense_sandwich_co(X: DenseDev, d, want_colsum=False, center=None):
    """X' diag(d) X of an unrestricted C-ordered float64 block of an even number of columns
    <= 128 with the kernel that is sized to share its compute units with a partner running on
    another stream (tm_dense_sandwich_co_f64; reference: the dense term of
    split_matrix.py:337-354).  Returns out, or (out, X' d) with want_colsum.  center (float64 device
    tensor, length X.m): product and column sums of X - 1 center' (tm_dense_sandwich_co_centered_f64)."""
    import torch

    out = D.out_buf((X.m, X.m), torch.float64)
    cs = D.out_buf((X.m,), torch.float64) if want_colsum else None
    D.same_float("dense_sandwich_co", X.buf, d, out)
    if center is not None:
        D.same_float("dense_sandwich_co", X.buf, center)
        assert center.numel() == X.m and center.is_contiguous()
        call("tm_dense_sandwich_co_centered_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(center), D.p(out), D.p(cs),
             D.stream_ptr())
        return (out, cs) if want_colsum else out
    call("tm_dense_sandwich_co_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(out), D.p(cs), D.stream_ptr())
    return (out, cs) if want_colsum else out


def co_supported(X: DenseDev, d, any_width=False) -> bool:
    """True when tm_dense_sandwich_co_f64 takes the block AND pays for it: the kernel always works on
    a 128-column panel, so blocks of <= 64 columns stay with the narrow syrk instantiations (plus a
    transpose_matvec where X'd is wanted) unless any_width is set."""
    import torch

    return (not X.order_f and X.buf.dtype == torch.float64 and d.dtype == torch.float64
            and X.m <= 128 and X.m > 0 and X.n > 0 and X.buf.data_ptr() % 16 == 0
            and (any_width or X.m > 64))


def dense_sandwich_bf16x3(X: DenseDev, d):
    """X' diag(d) X of an unrestricted C-ordered float32 block of 4 k <= 256 columns on the bf16
    matrix cores (tm_dense_sandwich_bf16x3_f32: three-piece bf16 split, f32 accumulation)."""
    import torch

    assert not X.order_f and X.buf.dtype == torch.float32 and X.m % 4 == 0 and X.m <= 256
    out = D.out_buf((X.m, X.m), torch.float32)
    D.same_float("dense_sandwich_bf16x3", X.buf, d)
    call("tm_dense_sandwich_bf16x3_f32", D.p(X.buf), X.n, X.m, D.p(d), D.p(out), D.stream_ptr())
    return out


def dense_sandwich_i8(X: DenseDev, d, colmax, want_colsum=False, history=None, center=None):
    """X' diag(d) X of an unrestricted C-ordered float64 block of an even number of columns <= 128 on
    the int8 matrix cores (tm_dense_sandwich_i8_f64: 40-bit fixed point per column, five base-256
    digits, 22 exact int8 digit-pair products).  colmax: float64 device tensor of max |x| per column.
    Weights outside the envelope (negative, non-finite, tiny exactly where a column is large) make
    the call run the f64 kernel instead (checked on the device).  want_colsum: also X' d from the
    same pass -> (out, colsum).  history: int32 device tensor of tm_dense_sandwich_i8_history_words() words
    kept per matrix (tm_dense_sandwich_i8_hist_f64: {misses in a row, calls, -, -, the previous diagonal as
    128 doubles}; after three misses in a row the int8 attempt is skipped).  center (float64 device tensor,
    length X.m): product and column sums of X - 1 center', the centre subtracted BEFORE the fixed-point
    conversion (tm_dense_sandwich_i8_centered_f64; colmax is then max |x - center| per column)."""
    import torch
    from .._lib import lib

    out = D.out_buf((X.m, X.m), torch.float64)
    D.same_float("dense_sandwich_i8", X.buf, d, colmax)
    cs = D.out_buf((X.m,), torch.float64) if want_colsum else None
    if history is not None and history.numel() < int(lib().tm_dense_sandwich_i8_history_words()):
        # (the C ABI takes no length: a short buffer would be a silent out-of-bounds device write)
        raise ValueError("history needs tm_dense_sandwich_i8_history_words() int32 words")
    if center is not None:
        D.same_float("dense_sandwich_i8", X.buf, center)
        assert center.numel() == X.m and center.is_contiguous()
        call("tm_dense_sandwich_i8_centered_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(center),
             D.p(out), D.p(cs), D.p(history), D.stream_ptr())
        return (out, cs) if want_colsum else out
    if history is not None:
        call("tm_dense_sandwich_i8_hist_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(out), D.p(cs),
             D.p(history), D.stream_ptr())
        return (out, cs) if want_colsum else out
    if want_colsum:
        call("tm_dense_sandwich_i8_xtd_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(out), D.p(cs),
             D.stream_ptr())
        return out, cs
    call("tm_dense_sandwich_i8_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(out), D.stream_ptr())
    return out


def dense_sandwich_i8_wide(X: DenseDev, d, colmax, center=None):
    """X' diag(d) X of an unrestricted C-ordered float64 block of 130 .. 512 (even) columns: the diagonal
    128-column panels on the int8 matrix cores in place, the off-diagonal panel pairs on the f64 MFMA
    (tm_dense_sandwich_i8_wide_f64; reference: the j-panels of ext/dense_helpers-tmpl.cpp:289)."""
    import torch

    out = D.out_buf((X.m, X.m), torch.float64)
    D.same_float("dense_sandwich_i8_wide", X.buf, d, colmax)
    if center is not None:
        D.same_float("dense_sandwich_i8_wide", X.buf, center)
        assert center.numel() == X.m and center.is_contiguous()
        call("tm_dense_sandwich_i8_wide_centered_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(center),
             D.p(out), D.stream_ptr())
        return out
    call("tm_dense_sandwich_i8_wide_f64", D.p(X.buf), X.n, X.m, D.p(d), D.p(colmax), D.p(out), D.stream_ptr())
    return out
