"""StandardizedMatrix: lazily centred / scaled view  self[i, j] = mult[j] * mat[i, j] + shift[j]
(reference: /root/reference/src/tabmat/standardized_mat.py) over device-resident blocks.

Everything stays on the device: shift / mult are kept in HBM next to their host copies, the
per-call vector may be a numpy array (numpy result) or a torch cuda tensor (device result, no
host traffic), the O(p^2) rank-one corrections of the sandwich run in
tm_standardize_sandwich_f64 and the sums over the per-call vector in tm_vec_sum_*.  The sandwich
gets X' d out of the SAME pass over the blocks as the inner sandwich wherever the algebra allows
it (SplitMatrix._sandwich_xtd_dev); the reference makes two passes (standardized_mat.py:148-150)."""
from __future__ import annotations

import numpy as np
import torch

from . import _device as D
from ._lib import call
from .matrix_base import MatrixBase
from .util import (
    check_matvec_dimensions,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    normalize_index,
    set_up_rows_or_cols,
    setup_restrictions,
)


def _vec_sum(v_dev, rows_d):
    """sum(v[rows]) as a float64 device scalar (tm_vec_sum_*)."""
    out = D.zeros((1,), torch.float64)
    n = v_dev.numel() if rows_d is None else D.nlen(rows_d)
    if n:
        call(f"tm_vec_sum_{D.fsuf(v_dev)}", D.p(v_dev), D.p(rows_d), n, D.p(out), D.stream_ptr())
    return out


class StandardizedMatrix:
    __array_priority__ = 11

    def __init__(self, mat: MatrixBase, shift, mult=None):
        if not isinstance(mat, MatrixBase):
            raise TypeError("mat should be an instance of a MatrixBase subclass.")
        if isinstance(shift, torch.Tensor):
            shift = D.to_host(shift)
        if isinstance(mult, torch.Tensor):
            mult = D.to_host(mult)
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
        self._dev_cache = {}

    # ---- device copies of the p-sized vectors ---------------------------------------------
    def _shift_dev(self, cols_n=None, dtype=torch.float64):
        return self._pvec("shift", self.shift, cols_n, dtype)

    def _mult_dev(self, cols_n=None, dtype=torch.float64):
        return None if self.mult is None else self._pvec("mult", self.mult, cols_n, dtype)

    def _pvec(self, name, arr, cols_n, dtype):
        key = (name, dtype)
        if key not in self._dev_cache:
            self._dev_cache[key] = D.to_dev(np.asarray(arr, dtype=np.float64), dtype)
        t = self._dev_cache[key]
        return t if cols_n is None else t[D.idx_dev(cols_n, torch.int64)]

    # ---- products -------------------------------------------------------------------------
    def matvec(self, other_mat, cols=None, out=None):
        """standardized_mat.py:69-97: mat.matvec(mult * v, cols) + shift[cols] . v[cols]."""
        on_dev = D.is_dev(other_mat)
        if on_dev and other_mat.ndim == 1:
            v = other_mat
            check_matvec_dimensions(self, v, transpose=False)
            cols_n = normalize_index(cols, self.shape[1])
            tdt = D.torch_dtype(self.dtype)
            v = D.to_dev(v, tdt)
            scaled = v if self.mult is None else v * self._mult_dev(None, tdt)
            sh = self._shift_dev(cols_n, tdt)
            vc = v if cols_n is None else v[D.idx_dev(cols_n, torch.int64)]
            # the scalar shift . v is the start value of the accumulating block kernels
            res = (sh * vc).sum().expand(self.shape[0]).contiguous()
            res = self.mat.matvec(scaled, cols, out=res)
            if out is None:
                return res
            out += res
            return out
        cols = set_up_rows_or_cols(cols, self.shape[1])
        other_mat = np.asarray(D.to_host(other_mat) if on_dev else other_mat)
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

    def _inner_xtd_dev(self, d, rows_d, cols_n):
        """(inner sandwich or None, its diagonal or None, mat' d) as float64 device tensors."""
        from .categorical_matrix import CategoricalMatrix
        from .split_matrix import SplitMatrix

        mat = self.mat
        if isinstance(mat, SplitMatrix):
            inner, xtd = mat._sandwich_xtd_dev(d, rows_d, cols_n)
            return inner, None, xtd
        cols_d = D.idx_dev(cols_n)
        if isinstance(mat, CategoricalMatrix):
            diag = mat._sandwich_diag_dev(d, rows_d, cols_d).to(torch.float64)
            return None, diag, diag           # one-hot entries are 0 / 1: C' d = diag(C' D C)
        from .dense_matrix import DenseMatrix
        from .ext import dense as xd

        if isinstance(mat, DenseMatrix) and rows_d is None and cols_d is None:
            both = mat._sandwich_xtd_dev(d)                                       # one pass
            if both is not None:
                return both[0], None, both[1]
        inner = mat._sandwich_dev(d, rows_d, cols_d).to(torch.float64)
        xtd = mat._matvec_dev(d, rows_d, cols_d, None, True).to(torch.float64)
        return inner, None, xtd

    def sandwich(self, d, rows=None, cols=None):
        """Inner sandwich + rank-one corrections (standardized_mat.py:123-172), float64.
        d: numpy array (numpy result) or torch cuda tensor (device result)."""
        on_dev = D.is_dev(d)
        if not on_dev and not hasattr(d, "dtype"):
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        rows_n = normalize_index(rows, self.shape[0])
        cols_n = normalize_index(cols, self.shape[1])
        d_dev = D.to_dev(d)
        rows_d = D.idx_dev(rows_n)
        k = self.shape[1] if cols_n is None else len(cols_n)
        if rows_n is not None and len(rows_n) == 0:
            res = D.zeros((k, k), torch.float64)
            return res if on_dev else D.to_host(res)
        inner, diag, xtd = self._inner_xtd_dev(d_dev, rows_d, cols_n)
        res = inner.contiguous() if inner is not None else D.empty((k, k), torch.float64)
        sum_d = _vec_sum(d_dev, rows_d)
        # (operands held in locals until the launch is queued: a temporary whose pointer has been
        # taken would hand its memory to the next temporary)
        xtd_c = xtd.contiguous()
        shift_c = self._shift_dev(cols_n).contiguous()
        mult_c = None if self.mult is None else self._mult_dev(cols_n).contiguous()
        call("tm_standardize_sandwich_f64", D.p(res), D.p(diag), D.p(xtd_c), D.p(shift_c),
             D.p(mult_c), D.p(sum_d), k, D.stream_ptr())
        return res if on_dev else D.to_host(res)

    def unstandardize(self) -> MatrixBase:
        return self.mat

    def transpose_matvec(self, other, rows=None, cols=None, out=None):
        """standardized_mat.py:178-230: mult[cols] * mat.T[cols, rows] other[rows]
        + shift[cols] * sum(other[rows])."""
        check_transpose_matvec_out_shape(self, out)
        on_dev = D.is_dev(other)
        if on_dev and other.ndim == 1:
            check_matvec_dimensions(self, other, transpose=True)
            rows_n = normalize_index(rows, self.shape[0])
            cols_n = normalize_index(cols, self.shape[1])
            tdt = D.torch_dtype(self.dtype)
            v = D.to_dev(other, tdt)
            res = self.mat.transpose_matvec(v, rows, cols)
            s = _vec_sum(v, D.idx_dev(rows_n)).to(tdt)
            if self.mult is not None:
                res = res * self._mult_dev(cols_n, tdt)
            res = res + self._shift_dev(cols_n, tdt) * s
            if out is None:
                return res
            if cols_n is None:
                out += res
            else:
                out[D.idx_dev(cols_n, torch.int64)] += res
            return out
        other = np.asarray(D.to_host(other) if on_dev else other)
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
