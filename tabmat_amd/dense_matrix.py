"""DenseMatrix: a C- or F-contiguous 2-D array resident in HBM (reference:
/root/reference/src/tabmat/dense_matrix.py).  sandwich -> MFMA row-weighted syrk,
matvec / transpose_matvec -> streaming gemv kernels (tabmat_amd/csrc/dense.hip)."""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from . import _device as D
from .ext import dense as xd
from .ext._types import DenseDev
from .matrix_base import MatrixBase
from .util import (
    check_indexer,
    check_matvec_dimensions,
    check_matvec_out_shape,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    normalize_index,
    np_dtype_of,
    device_row_index,
    selects_all_columns,
)


# F-ordered blocks get a row-major twin in HBM for the sandwich kernels (DenseMatrix._dev_c);
# TABMAT_AMD_ROW_MAJOR_TWIN=0 keeps the caller's layout only (column-major kernel variants).
ROW_MAJOR_TWIN = os.environ.get("TABMAT_AMD_ROW_MAJOR_TWIN", "1") != "0"
# the float64 syrk on the int8 matrix cores (Ozaki-style slicing, csrc/syrk_i8.hip) for blocks it suits
SYRK_I8 = os.environ.get("TABMAT_AMD_SYRK_I8", "1") != "0"
I8_MIN_ROWS = 4096
I8_MAX_COLS = 512      # above 128 columns: 128-column panels (diagonal panels int8, panel pairs f64 MFMA)
# a row restriction that keeps at least this share of the rows runs the unrestricted int8 syrk on a masked d
I8_MASKED_ROWS_SHARE = 0.25


def set_strict_f64(flag: bool = True) -> bool:
    """strict = True: every float64 sandwich runs on the float64 MFMA / vector units.  By default the dense
    self term of a qualifying block (C-ordered, 65..128 or 130..512 even columns, >= 4096 rows, finite) is computed in
    40-bit fixed point per column on the int8 matrix cores (K1e, csrc/syrk_i8.hip): entry-wise error below
    1e-10 * sqrt(S_ii S_jj) by its on-device envelope check (observed 2e-14 of max|S|), the bar BASELINE.json
    sets -- but not IEEE float64 to the last bits.  (StandardizedMatrix.sandwich hands the kernels the column
    centres, so the fixed point is spent on x - mean and nothing is amplified by (mean / std)^2: round 5.)
    Takes effect at the next call (no cache to clear); returns the previous setting.
    TABMAT_AMD_SYRK_I8=0 sets strict mode at import."""
    global SYRK_I8
    old = not SYRK_I8
    SYRK_I8 = not flag
    return old


def strict_f64() -> bool:
    return not SYRK_I8


_STRICT_F32 = os.environ.get("TABMAT_AMD_SYRK_BF16", "1") == "0"


def set_strict_f32(flag: bool = True) -> bool:
    """strict = True: every float32 sandwich runs on the float32 MFMA.  By default an unrestricted C-ordered
    float32 block of 129..256 columns is computed as a three-piece bf16 split on the bf16 matrix cores (K1d,
    csrc/syrk_bf16.hip: 3.5 instead of 5.8 ms at BASELINE configs[1]) with a relative error of 6.6e-7 of
    max|S| against 8e-8 for the f32 MFMA -- inside the reference's own float32 tolerance (sqrt(eps),
    tests/test_fast_sandwich.py), but not the last bits of float32.  The counterpart of set_strict_f64; sets
    the library's `syrk_bf16` knob.  TABMAT_AMD_SYRK_BF16=0 sets strict mode at import.  Returns the previous
    setting."""
    global _STRICT_F32
    from ._lib import call

    old = _STRICT_F32
    _STRICT_F32 = bool(flag)
    call("tm_tune_set", b"syrk_bf16", 0 if flag else 1)
    return old


def strict_f32() -> bool:
    return _STRICT_F32


class DenseMatrix(MatrixBase):
    """Dense block.  Construct from a numpy array (kept on the host, uploaded lazily on the
    first product) or from an (n, m) torch cuda tensor (no host copy)."""

    def __init__(self, input_array, column_names=None, term_names=None):
        self._devblk = None
        if isinstance(input_array, torch.Tensor):
            t = input_array.reshape(-1, 1) if input_array.ndim == 1 else input_array
            if t.ndim != 2:
                raise ValueError("Input array must be 1- or 2-dimensional")
            self._array = None
            self._devblk = DenseDev.from_tensor(D.to_dev(t) if not t.is_cuda else t)
            self._shape = tuple(t.shape)
            self._dtype = np_dtype_of(t)
        else:
            a = np.asarray(input_array)
            if a.ndim == 1:
                a = a.reshape(-1, 1)
            elif a.ndim > 2:
                raise ValueError("Input array must be 1- or 2-dimensional")
            if not (a.flags["C_CONTIGUOUS"] or a.flags["F_CONTIGUOUS"]):
                # dense_matrix.py:47-58
                warnings.warn("Input array is not contiguous; making a copy.", UserWarning,
                              stacklevel=2)
                a = np.asfortranarray(a)
            self._array = a
            self._shape = a.shape
            self._dtype = a.dtype
        width = self._shape[1]
        if column_names is not None and len(column_names) != width:
            raise ValueError(f"Expected {width} column names, got {len(column_names)}")
        if term_names is not None and len(term_names) != width:
            raise ValueError(f"Expected {width} term names, got {len(term_names)}")
        self._colnames = list(column_names) if column_names is not None else [None] * width
        self._terms = list(term_names) if term_names is not None else self._colnames

    # ---- storage ------------------------------------------------------------------------
    def _dev(self) -> DenseDev:
        if self._devblk is None:
            self._devblk = DenseDev.from_host(self._array)
        return self._devblk

    def _dev_c(self) -> DenseDev:
        """C-ordered (row-major) device view for the sandwich kernels.  An F-ordered block gets a
        row-major TWIN in HBM on first use (the kernels that stream rows -- MFMA syrk, the gather
        K3, categorical x dense -- are built around 16-byte row segments; the reference likewise
        keeps a CSR twin next to its CSC arrays, sparse_matrix.py:133-143).  matvec /
        transpose_matvec keep using the caller's layout.  When HBM is short the native layout is
        returned and the (slower) column-major kernel variants run."""
        blk = self._dev()
        if not blk.order_f or not ROW_MAJOR_TWIN:
            return blk
        twin = getattr(self, "_devblk_c", None)
        if twin is None:
            need = blk.buf.numel() * blk.buf.element_size()
            free = torch.cuda.mem_get_info(blk.buf.device)[0] if blk.buf.is_cuda else 0
            # decided once (no memory queries inside a captured launch sequence later)
            twin = blk if free < 2 * need + (1 << 30) else \
                DenseDev(blk.as_2d().contiguous(), blk.n, blk.m, 0)
            self._devblk_c = twin
        return twin

    def _values_finite(self) -> bool:
        """True when the block holds no inf / nan (checked once; see SplitMatrix.matvec)."""
        if getattr(self, "_finite", None) is None:
            self._finite = bool(torch.isfinite(self._dev().buf).all().item())
        return self._finite

    def to_device(self):
        """Upload now (otherwise the first product does it)."""
        self._dev_c()
        return self

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def ndim(self):
        return 2

    __array_ufunc__ = None

    def toarray(self) -> np.ndarray:
        if self._array is None:
            self._array = D.to_host(self._devblk.as_2d())
        return self._array

    def unpack(self):
        return self.toarray()

    def transpose(self):
        return type(self)(self.toarray().T)

    T = property(transpose)

    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        return type(self)(self.toarray().astype(dtype, order, casting, copy),
                          column_names=self._colnames, term_names=self._terms)

    def getcol(self, i):
        return type(self)(self.toarray()[:, [i]], column_names=[self._colnames[i]],
                          term_names=[self._terms[i]])

    def __getitem__(self, key):
        row, col = check_indexer(key)
        if self._devblk is not None and selects_all_columns(col, self.shape[1]):
            # row indexing of a block that lives in HBM stays in HBM (no host round trip)
            kind, *arg = device_row_index(row, self.shape[0])
            t = self._devblk.as_2d()
            sub = t[arg[0]:arg[1]] if kind == "slice" else t[D.idx_dev(arg[0], torch.int64)]
            return type(self)(sub, column_names=self._colnames, term_names=self._terms)
        names = np.array(self._colnames, dtype=object)[col].ravel().tolist()
        terms = np.array(self._terms, dtype=object)[col].ravel().tolist()
        return type(self)(self.toarray()[row, col], column_names=names, term_names=terms)

    def __matmul__(self, other):
        return self.matvec(other)

    def __str__(self):
        return "{}x{} DenseMatrix:\n\n".format(*self.shape) + np.array_str(self.toarray())

    def __repr__(self):
        return f"DenseMatrix({np.array2string(self.toarray(), separator=', ')})"

    def multiply(self, other):
        other = np.asanyarray(other)
        arr = self.toarray() * (other[:, None] if other.ndim == 1 else other)
        return type(self)(arr, column_names=self._colnames, term_names=self._terms)

    # ---- hot path -----------------------------------------------------------------------
    def _i8_colmax(self, center=None):
        """max |x| per column (float64 device tensor) when the block qualifies for the int8-sliced
        syrk (csrc/syrk_i8.hip), else None: finite float64 blocks of 65 .. 128 columns or 130 .. 512 even ones.  One
        pass over the block at first use (column maxima and minima are kept); with `center` the result is
        max |x - center| per column (the envelope of the centred fixed point).  The part of the envelope that
        depends on the weights (negative / non-finite d, weights tiny exactly where a column is large) is
        checked on the device inside every call, which then runs the f64 kernel instead."""
        hit = getattr(self, "_i8_ok", None)
        if hit is None:
            hit = False
            blk = self._dev_c()
            # (odd widths up to 128 columns since round 5; the 128-column panels of wider blocks need even ones)
            if (not blk.order_f and blk.buf.dtype == torch.float64 and 64 < blk.m <= I8_MAX_COLS
                    and (blk.m % 2 == 0 or blk.m <= 128) and blk.n >= I8_MIN_ROWS
                    and blk.buf.data_ptr() % 16 == 0):
                # (two reductions, no |X| copy of the block: it is 10 GB at BASELINE configs[3]; a NaN
                # propagates through amax / amin, +-inf shows in one of them)
                x = blk.as_2d()
                hi, lo = x.amax(dim=0), x.amin(dim=0)
                if bool((torch.isfinite(hi) & torch.isfinite(lo)).all().item()):
                    hit = (hi, lo, torch.maximum(hi, -lo).contiguous())
            self._i8_ok = hit
        if hit is False:
            return None
        if center is None:
            return hit[2]
        return torch.maximum(hit[0] - center, center - hit[1]).contiguous()

    def _i8_history(self, key=None):
        """int32 words on the device: {consecutive envelope misses, calls, -, -, the previous call's diagonal
        (128 doubles)} of this block's int8 syrk (tm_dense_sandwich_i8_hist_f64).  One history per kind of
        call (`key`: None = plain, "masked" = row-masked weights, "centered" = centred columns): the
        previous-diagonal prediction compares like with like."""
        hs = self.__dict__.setdefault("_i8_hist", {})
        h = hs.get(key)
        if h is None:
            from ._lib import lib

            h = hs[key] = torch.zeros(int(lib().tm_dense_sandwich_i8_history_words()), dtype=torch.int32,
                                      device=self._dev_c().buf.device)
        return h

    def _center_dev(self, center):
        """`center` (length m: host array, device tensor or None) as a contiguous device tensor of the block's
        dtype (None stays None)."""
        if center is None:
            return None
        tdt = D.torch_dtype(self.dtype)
        c = D.to_dev(center, tdt) if not isinstance(center, torch.Tensor) else center.to(tdt)
        return c.contiguous()

    def _sandwich_dev(self, d, rows, cols, center=None):
        """center (device tensor over ALL columns of the block, block dtype, or None): the product of
        X - 1 center' -- every dense syrk subtracts the centre on the way in (csrc/dense.hip, syrk_co.hip,
        syrk_i8.hip); used by StandardizedMatrix.sandwich."""
        if (SYRK_I8 and cols is None and d.dtype == torch.float64
                and (rows is None or D.nlen(rows) >= I8_MASKED_ROWS_SHARE * self.shape[0])):
            cmax = self._i8_colmax(center)
            if cmax is not None:
                key = None if center is None else "centered"
                if rows is not None:
                    # excluded rows get d = 0 (the block is finite: checked once in _i8_colmax), one full pass;
                    # a row id that occurs twice counts twice (D.masked_d), as in the reference's row loop
                    # (dense_helpers-tmpl.cpp:224) and in the row-list kernels below the threshold
                    d = D.masked_d(d, rows)
                    key = "masked" if center is None else "masked-centered"
                if self.shape[1] > 128:
                    return xd.dense_sandwich_i8_wide(self._dev_c(), d, cmax, center=center)
                return xd.dense_sandwich_i8(self._dev_c(), d, cmax, history=self._i8_history(key), center=center)
        return xd.dense_sandwich(self._dev_c(), d, rows, cols, center=center)

    def _sandwich_xtd_dev(self, d, center=None):
        """(X' diag(d) X, X' d) of the unrestricted block in ONE pass over it, or None when no
        one-pass kernel takes the block (then the caller makes the reference's second pass,
        standardized_mat.py:149-150): the int8-sliced syrk (K1e) inside its envelope, else the f64
        syrk with the column sums of its A-side fragments (K1c).  With `center` both are those of
        X - 1 center'."""
        blk = self._dev_c()
        if SYRK_I8 and d.dtype == torch.float64 and blk.m <= 128:
            cmax = self._i8_colmax(center)
            if cmax is not None:
                return xd.dense_sandwich_i8(blk, d, cmax, want_colsum=True,
                                            history=self._i8_history(None if center is None else "centered"),
                                            center=center)
        if xd.co_supported(blk, d):
            return xd.dense_sandwich_co(blk, d, want_colsum=True, center=center)
        return None

    def sandwich(self, d, rows=None, cols=None):
        """X[rows, cols].T @ diag(d[rows]) @ X[rows, cols] (dense_matrix.py:153-163)."""
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        rows = normalize_index(rows, self.shape[0])
        cols = normalize_index(cols, self.shape[1])
        res = self._sandwich_dev(D.to_dev(d), D.idx_dev(rows), D.idx_dev(cols))
        return res if on_dev else D.to_host(res)

    def _cross_sandwich_dev(self, other, d, rows, L_cols, R_cols):
        from .categorical_matrix import CategoricalMatrix
        from .sparse_matrix import SparseMatrix

        if isinstance(other, (SparseMatrix, CategoricalMatrix)):
            return other._cross_sandwich_dev(self, d, rows, R_cols, L_cols).T
        raise TypeError

    def _cross_sandwich(self, other, d, rows=None, L_cols=None, R_cols=None):
        """dense_matrix.py:165-178."""
        on_dev = D.is_dev(d)
        res = self._cross_sandwich_dev(
            other, D.to_dev(d if on_dev else np.asarray(d)), D.idx_dev(normalize_index(rows, self.shape[0])),
            D.idx_dev(normalize_index(L_cols, self.shape[1])),
            D.idx_dev(normalize_index(R_cols, other.shape[1])))
        return res if on_dev else D.to_host(res)

    def _get_col_stds(self, weights, col_means):
        """sqrt(sum_i w_i (x_ij - mean_j)^2) (dense_matrix.py:180-187) with the K7 kernel
        (transpose_square_dot_weights, ext/dense.pyx:103-122)."""
        tdt = D.torch_dtype(self.dtype)
        arg = D.to_host(xd.transpose_square_dot_weights(
            self._dev(), D.to_dev(np.asarray(weights), tdt), D.to_dev(np.asarray(col_means), tdt)))
        arg[arg < 0] = 0
        return np.sqrt(arg)

    def _matvec_dev(self, vec, rows, cols, out, transpose):
        """vec / out are 1-D device tensors; accumulates into out (created when None)."""
        fn = xd.dense_rmatvec if transpose else xd.dense_matvec
        return fn(self._dev(), vec, rows, cols, out)

    def _matvec_helper(self, vec, rows, cols, out, transpose):
        on_dev = D.is_dev(vec)
        if not on_dev:
            vec = np.asarray(vec)
        check_matvec_dimensions(self, vec, transpose=transpose)
        n, m = self.shape
        rows_n = normalize_index(rows, n)
        cols_n = normalize_index(cols, m)
        # "we assume that rows and cols are unique" (dense_matrix.py:208-210)
        if rows_n is not None and len(rows_n) == n:
            rows_n = None
        if cols_n is not None and len(cols_n) == m:
            cols_n = None
        tdt = D.torch_dtype(self.dtype)
        v_dev = D.to_dev(vec, tdt)
        rd, cd = D.idx_dev(rows_n), D.idx_dev(cols_n)
        if v_dev.ndim == 1:
            res = self._matvec_dev(v_dev, rd, cd, None, transpose)
        else:
            res = xd.dense_matvec_multi(self._dev(), v_dev, rd, cd, transpose)
        if not on_dev:
            res = D.to_host(res)
            if np.issubdtype(vec.dtype, np.floating) and vec.dtype != self.dtype:
                res = res.astype(np.result_type(vec.dtype, self.dtype))
        if out is None:
            return res
        if transpose and cols_n is not None:
            out[cols_n if not D.is_dev(out) else D.idx_dev(cols_n, torch.int64)] += res
        else:
            out += res
        return out

    def transpose_matvec(self, vec, rows=None, cols=None, out=None):
        """self[rows, cols].T @ vec[rows] (dense_matrix.py:238-247)."""
        check_transpose_matvec_out_shape(self, out)
        return self._matvec_helper(vec, rows, cols, out, True)

    def matvec(self, vec, cols=None, out=None):
        """self[:, cols] @ vec[cols] (dense_matrix.py:249-257)."""
        check_matvec_out_shape(self, out)
        return self._matvec_helper(vec, None, cols, out, False)


if _STRICT_F32:        # TABMAT_AMD_SYRK_BF16=0: the knob lives in the library
    try:
        set_strict_f32(True)
    except Exception:   # library not built yet: every product will raise TabmatHipError anyway
        pass
