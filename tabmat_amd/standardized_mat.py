"""StandardizedMatrix: lazily centred / scaled view  self[i, j] = mult[j] * mat[i, j] + shift[j]
(reference: /root/reference/src/tabmat/standardized_mat.py).  Everything heavy is delegated to
the wrapped matrix' device products; the corrections are O(p^2) host arithmetic."""
from __future__ import annotations

import numpy as np
from scipy import sparse as sps

from .matrix_base import MatrixBase
from .util import (
    check_matvec_dimensions,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    set_up_rows_or_cols,
    setup_restrictions,
)


class StandardizedMatrix:
    __array_priority__ = 11

    def __init__(self, mat: MatrixBase, shift, mult=None):
        if not isinstance(mat, MatrixBase):
            raise TypeError("mat should be an instance of a MatrixBase subclass.")
        shift_arr = np.atleast_1d(np.squeeze(shift))
        want = (mat.shape[1],)
        if shift_arr.shape != want:
            raise ValueError(f"Expected shift to be able to conform to shape {want}, "
                             f"but it has shape {np.asarray(shift).shape}")
        mult_arr = None
        if mult is not None:
            mult_arr = np.atleast_1d(np.squeeze(mult))
            if mult_arr.shape != want:
                raise ValueError(f"Expected mult to be able to conform to shape {want}, "
                                 f"but it has shape {np.asarray(mult).shape}")
        self.shift, self.mult, self.mat = shift_arr, mult_arr, mat
        self.shape, self.ndim, self.dtype = mat.shape, mat.ndim, mat.dtype

    def matvec(self, other_mat, cols=None, out=None):
        """standardized_mat.py:69-97."""
        cols = set_up_rows_or_cols(cols, self.shape[1])
        other_mat = np.asarray(other_mat)
        check_matvec_dimensions(self, other_mat, transpose=False)
        scaled = other_mat
        if self.mult is not None:
            scaled = self.mult.reshape((-1,) + (1,) * (other_mat.ndim - 1)) * other_mat
        res = self.mat.matvec(scaled, cols, out=out)
        res += self.shift[cols].dot(other_mat[cols, ...])
        return res

    def getcol(self, i: int):
        mult = None if self.mult is None else [self.mult[i]]
        return StandardizedMatrix(self.mat.getcol(i), [self.shift[i]], mult)

    def sandwich(self, d, rows=None, cols=None):
        """Inner sandwich + rank-one corrections (standardized_mat.py:123-172)."""
        if not hasattr(d, "dtype"):
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        if rows is not None or cols is not None:
            r_, c_ = setup_restrictions(self.shape, rows, cols)
            rows = r_ if rows is not None else None
            cols = c_ if cols is not None else None
        inner = self.mat.sandwich(d, rows, cols)
        d_mat = self.mat.transpose_matvec(d, rows, cols)
        lim_mult = None
        if self.mult is not None:
            lim_mult = self.mult[cols] if cols is not None else self.mult
            d_mat = d_mat * lim_mult
        lim_shift = self.shift[cols] if cols is not None else self.shift
        lim_d = d[rows] if rows is not None else d
        res = (np.outer(d_mat, lim_shift) + np.outer(lim_shift, d_mat)
               + np.outer(lim_shift, lim_shift) * np.sum(lim_d))
        if sps.issparse(inner):
            diag = np.asarray(inner.diagonal(), dtype=float)
            if lim_mult is not None:
                diag = diag * lim_mult**2
            k = np.arange(res.shape[0])
            res[k, k] += diag
        else:
            res += inner * np.outer(lim_mult, lim_mult) if lim_mult is not None else inner
        return res

    def unstandardize(self) -> MatrixBase:
        return self.mat

    def transpose_matvec(self, other, rows=None, cols=None, out=None):
        """standardized_mat.py:178-230."""
        check_transpose_matvec_out_shape(self, out)
        other = np.asarray(other)
        check_matvec_dimensions(self, other, transpose=True)
        res = self.mat.transpose_matvec(other, rows, cols)
        rows_a, cols_a = setup_restrictions(self.shape, rows, cols)
        other_sum = np.sum(other[rows_a], 0)
        shift_part = np.reshape(np.outer(self.shift[cols_a], other_sum),
                                (len(cols_a),) + res.shape[1:])
        if self.mult is not None:
            res = res * self.mult[cols_a].reshape((-1,) + (1,) * (res.ndim - 1))
        res = res + shift_part
        if out is None:
            return res
        out[cols_a] += res
        return out

    def __rmatmul__(self, other):
        if not hasattr(other, "T"):
            other = np.asarray(other)
        return self.transpose_matvec(other.T).T

    def __matmul__(self, other):
        return self.matvec(other)

    def toarray(self) -> np.ndarray:
        base = self.mat.toarray()
        if self.mult is not None:
            base = self.mult[None, :] * base
        return base + self.shift[None, :]

    @property
    def A(self):
        return self.toarray()

    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        return type(self)(self.mat.astype(dtype, casting=casting, copy=copy), self.shift, self.mult)

    def __getitem__(self, item):
        if isinstance(item, tuple):
            row, col = item
        else:
            row, col = item, slice(None)
        mult = None if self.mult is None else np.atleast_1d(self.mult[col])
        return StandardizedMatrix(self.mat[row, col] if not (isinstance(col, slice) and col == slice(None))
                                  else self.mat[row, :], np.atleast_1d(self.shift[col]), mult)
