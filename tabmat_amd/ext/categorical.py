"""Mirror of tabmat.ext.categorical (reference: src/tabmat/ext/categorical.pyx).  The
fast/complex split of the reference (drop_first / missing handling) is a runtime flag here."""
from __future__ import annotations

from .. import _device as D
from .._lib import call


def transpose_matvec(indices, other, n_cols, rows, cols, out, drop_first=False):
    """ext/categorical.pyx:23-117 (transpose_matvec_fast/_complex): out[c] += ... in place;
    `out` has full block width n_cols, only columns in `cols` are touched."""
    n = indices.numel()
    if (rows is not None and D.nlen(rows) == 0) or (cols is not None and D.nlen(cols) == 0):
        return  # an empty tensor has a NULL data pointer, which the C ABI reads as "all"
    if rows is not None and D.nlen(rows) == n:
        rows = None
    if cols is not None and D.nlen(cols) == n_cols:
        cols = None
    D.same_float("transpose_matvec", other, out)
    call(f"tm_cat_transpose_matvec_{D.fsuf(out)}", D.p(indices), n, n_cols, int(drop_first),
         D.p(other), D.p(rows), D.nlen(rows), D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())


def transpose_matvec_det(perm, bstart, n_blocks, cat_bptr, n_cols, other, out, accumulate=True):
    """Deterministic form of transpose_matvec_fast/_complex (ext/categorical.pyx:23-117): fixed
    summation order per column (tm_cat_transpose_matvec_det_*)."""
    D.same_float("transpose_matvec_det", other, out)
    call(f"tm_cat_transpose_matvec_det_{D.fsuf(out)}", D.p(perm), D.p(bstart), int(n_blocks),
         D.p(cat_bptr), int(n_cols), D.p(other), D.p(out), int(bool(accumulate)), D.stream_ptr())


def matvec(indices, other, n_rows, cols, n_cols, out_vec, drop_first=False):
    """ext/categorical.pyx:128-180 (matvec_fast/_complex): out_vec[i] += other[col(i)]."""
    if cols is not None and D.nlen(cols) == 0:
        return
    D.same_float("matvec", other, out_vec)
    call(f"tm_cat_matvec_{D.fsuf(out_vec)}", D.p(indices), n_rows, n_cols, int(drop_first),
         D.p(other), D.p(cols), D.nlen(cols), D.p(out_vec), D.stream_ptr())


def matvec_assign(indices, other, n_rows, cols, n_cols, out_vec, drop_first=False):
    """The same into fresh storage (out_vec need not be initialised): out_vec[i] = other[col(i)] or 0."""
    if cols is not None and D.nlen(cols) == 0:      # (an empty tensor has no pointer to tell it from "all columns")
        out_vec.zero_()
        return
    D.same_float("matvec", other, out_vec)
    call(f"tm_cat_matvec_assign_{D.fsuf(out_vec)}", D.p(indices), n_rows, n_cols, int(drop_first),
         D.p(other), D.p(cols), D.nlen(cols), D.p(out_vec), D.stream_ptr())


def sandwich_categorical(indices, d, rows, n_cols, drop_first=False):
    """ext/categorical.pyx:183-218 (sandwich_categorical_fast/_complex): the diagonal."""
    res = D.zeros((n_cols,), d.dtype)
    if n_cols == 0 or (rows is not None and D.nlen(rows) == 0):
        return res
    call(f"tm_cat_transpose_matvec_{D.fsuf(d)}", D.p(indices), indices.numel(), n_cols,
         int(drop_first), D.p(d), D.p(rows), D.nlen(rows), None, 0, D.p(res), D.stream_ptr())
    return res
