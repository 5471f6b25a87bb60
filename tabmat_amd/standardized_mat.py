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

    # Centred inner products (round 5).  self[:, j] = mult_j x_j + shift_j = mult_j (x_j - c_j) + delta_j with
    # c_j = -shift_j / mult_j: for a standardized column c_j is its mean and delta_j is rounding-sized.  The
    # reference (standardized_mat.py:148-171) forms the RAW product X' D X and subtracts mean-sized rank-one
    # terms, which amplifies whatever error the product has by (mean / std)^2 -- 1.6e5 for a "year" column,
    # 1e8 for an id-like one; with the int8-sliced dense term (2e-14 of the raw scale) that is outside
    # BASELINE's 1e-10.  Here the dense kernels take the centres and compute (X - 1 c')' D (X - 1 c') directly
    # (the fixed point / the f64 sums see x - c), and only the cross terms with sparse / categorical blocks
    # carry first-order (mean / std) terms.
    CENTER_DENSE = True

    def _centering(self):
        """(c over all p columns as a float64 device tensor, group int32 device tensor [-1 = not centred],
        {block: centre over the block's columns}) or None when there is nothing to centre: float64 matrices
        only (the contract is the float64 1e-10), dense blocks only (sparse / categorical columns stay as they
        are: centring would densify them)."""
        hit = self._dev_cache.get("centering", False)
        if hit is not False:
            return hit
        from .dense_matrix import DenseMatrix
        from .split_matrix import SplitMatrix

        res = None
        mat = self.mat
        if self.CENTER_DENSE and np.dtype(self.dtype) == np.float64:
            if isinstance(mat, SplitMatrix):
                blocks = [(b, mb, idx) for b, (mb, idx) in enumerate(zip(mat.matrices, mat.indices))
                          if isinstance(mb, DenseMatrix) and mb.dtype == np.float64]
            elif isinstance(mat, DenseMatrix):
                blocks = [(0, mat, np.arange(self.shape[1]))]
            else:
                blocks = []
            p = self.shape[1]
            c = np.zeros(p)
            grp = np.full(p, -1, dtype=np.int32)
            with np.errstate(all="ignore"):
                c_all = -self.shift if self.mult is None else -self.shift / self.mult
            c_all = np.where(np.isfinite(c_all), c_all, 0.0)
            vec = {}
            for b, mb, idx in blocks:
                cb = c_all[idx]
                if np.any(cb != 0.0):
                    c[idx] = cb
                    grp[idx] = b
                    vec[b] = D.to_dev(np.ascontiguousarray(cb), torch.float64)
            if vec:
                res = (D.to_dev(c, torch.float64), D.to_dev(grp), vec)
        self._dev_cache["centering"] = res
        return res

    def _inner_xtd_dev(self, d, rows_d, cols_n, cen=None):
        """(inner sandwich or None, its diagonal or None, mat' d, centred-mask or None, groups or None) as device
        tensors.  cen: the result of _centering(): the dense self terms come out centred, the mask says
        which entries of mat' d are centred column sums, groups (when not None) replaces the per-block group
        vector (split_matrix._Centering)."""
        from .categorical_matrix import CategoricalMatrix
        from .split_matrix import SplitMatrix, _Centering

        mat = self.mat
        if isinstance(mat, SplitMatrix):
            if cen is not None:
                inner, xtd, cmask, grp = mat._sandwich_xtd_dev(d, rows_d, cols_n, center=_Centering(cen[2]))
                return inner, None, xtd, cmask, grp
            inner, xtd = mat._sandwich_xtd_dev(d, rows_d, cols_n)
            return inner, None, xtd, None, None
        cols_d = D.idx_dev(cols_n)
        if isinstance(mat, CategoricalMatrix):
            diag = mat._sandwich_diag_dev(d, rows_d, cols_d).to(torch.float64)
            return None, diag, diag, None, None     # one-hot entries are 0 / 1: C' d = diag(C' D C)
        from .dense_matrix import DenseMatrix

        cvec = None if cen is None else cen[2][0]
        if isinstance(mat, DenseMatrix) and rows_d is None and cols_d is None:
            both = mat._sandwich_xtd_dev(d, cvec)                                  # one pass
            if both is not None:
                cm = None if cvec is None else torch.ones_like(both[1], dtype=torch.bool)
                return both[0], None, both[1], cm, None
        if isinstance(mat, DenseMatrix) and cvec is not None:
            inner = mat._sandwich_dev(d, rows_d, cols_d, center=cvec).to(torch.float64)
        else:
            inner = mat._sandwich_dev(d, rows_d, cols_d).to(torch.float64)
        xtd = mat._matvec_dev(d, rows_d, cols_d, None, True).to(torch.float64)
        cm = None if cvec is None else torch.zeros_like(xtd, dtype=torch.bool)
        return inner, None, xtd, cm, None

    def sandwich(self, d, rows=None, cols=None):
        """Inner sandwich + rank-one corrections (standardized_mat.py:123-172), float64.
        d: numpy array (numpy result) or torch cuda tensor (device result).
        float64 matrices with dense blocks: the dense self terms are computed CENTRED (see _centering),
        so the result keeps the accuracy of the product at the scale of the standardized columns."""
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
        cen = self._centering() if d_dev.dtype == torch.float64 else None
        inner, diag, xtd, cmask, grp = self._inner_xtd_dev(d_dev, rows_d, cols_n, cen)
        res = inner.contiguous() if inner is not None else D.empty((k, k), torch.float64)
        sum_d = _vec_sum(d_dev, rows_d)
        # (operands held in locals until the launch is queued: a temporary whose pointer has been
        # taken would hand its memory to the next temporary)
        xtd_c = xtd.contiguous()
        shift_c = self._shift_dev(cols_n).contiguous()
        mult_c = None if self.mult is None else self._mult_dev(cols_n).contiguous()
        if cen is not None and diag is None:
            sel = None if cols_n is None else D.idx_dev(cols_n, torch.int64)
            c_c = (cen[0] if sel is None else cen[0][sel]).contiguous()
            g_c = (grp if grp is not None else cen[1] if sel is None else cen[1][sel]).contiguous()
            # raw column sums of centred columns -> centred: X' d - c sum(d)
            xtd_c = torch.where(cmask, xtd_c, xtd_c - c_c * sum_d).contiguous()
            call("tm_standardize_sandwich_centered_f64", D.p(res), D.p(xtd_c), D.p(c_c), D.p(g_c), D.p(shift_c),
                 D.p(mult_c), D.p(sum_d), k, D.stream_ptr())
            return res if on_dev else D.to_host(res)
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

    def multiply(self, other):
        """Element-wise multiplication with a vector of length n (standardized_mat.py:255-262): always a
        DenseMatrix, as in the reference (host-side convenience, not a hot-path product)."""
        from .dense_matrix import DenseMatrix

        return DenseMatrix(self.toarray()).multiply(other)

    def __repr__(self):
        return (f"StandardizedMat. Mat: {type(self.mat)} of shape {self.mat.shape}.\n"
                f"        Shift: {self.shift}\n        Mult: {self.mult}\n        ")

    # ---- names: those of the wrapped matrix (standardized_mat.py:311-378)
    def get_names(self, type="column", missing_prefix=None, indices=None):
        return self.mat.get_names(type, missing_prefix, indices)

    def set_names(self, names, type="column"):
        self.mat.set_names(names, type)

    @property
    def column_names(self):
        return self.get_names(type="column")

    @column_names.setter
    def column_names(self, names):
        self.set_names(names, type="column")

    @property
    def term_names(self):
        return self.get_names(type="term")

    @term_names.setter
    def term_names(self, names):
        self.set_names(names, type="term")

    def __getitem__(self, item):
        if isinstance(item, tuple):
            row, col = item
        else:
            row, col = item, slice(None)
        mult = None if self.mult is None else np.atleast_1d(self.mult[col])
        return StandardizedMatrix(self.mat[row, col] if not (isinstance(col, slice) and col == slice(None))
                                  else self.mat[row, :], np.atleast_1d(self.shift[col]), mult)
