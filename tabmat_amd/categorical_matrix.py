"""CategoricalMatrix: a one-hot encoded categorical column stored as int32 codes in HBM
(reference: /root/reference/src/tabmat/categorical_matrix.py).  code -1 = missing (all-zero
row, cat_missing_method="zero"); drop_first removes the column of code 0.  Kernels:
tabmat_amd/csrc/cat.hip (LDS-privatised weighted histograms)."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
from scipy import sparse as sps

from . import _device as D
from .dense_matrix import DenseMatrix
from .ext import categorical as xc
from .ext import split as xsplit
from .matrix_base import MatrixBase
from .sparse_matrix import SparseMatrix
from .util import (
    check_indexer,
    check_matvec_dimensions,
    check_matvec_out_shape,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    normalize_index,
    device_row_index,
)


# TABMAT_AMD_DETERMINISTIC=1 (or tabmat_amd.categorical_matrix.DETERMINISTIC = True): categorical
# transpose_matvec / sandwich diagonals are summed in a fixed order (bitwise reproducible, as the
# reference's K4a is) instead of with LDS float atomics; ~0.1 ms per product at 10M rows.
DETERMINISTIC = os.environ.get("TABMAT_AMD_DETERMINISTIC", "0") == "1"


def _factorize(x: np.ndarray):
    """Sorted-unique factorisation with missing (None / NaN) -> -1; the numpy-only path of
    categorical_matrix.py:224-230.  (DataFrame-library extraction is ingest, out of scope.)"""
    x = np.asarray(x)
    if x.dtype.kind in "OUS":
        obj = x.astype(object)
        na = np.array([v is None or (isinstance(v, float) and v != v) for v in obj])
    elif x.dtype.kind == "f":
        na = np.isnan(x)
    else:
        na = np.zeros(x.shape, dtype=bool)
    cats, inv = np.unique(x[~na], return_inverse=True)
    codes = np.full(x.shape, -1, dtype=np.int32)
    codes[~na] = inv
    return codes, cats


class CategoricalMatrix(MatrixBase):
    def __init__(self, cat_vec, categories: Optional[np.ndarray] = None, drop_first: bool = False,
                 dtype=np.float64, column_name: Optional[str] = None,
                 term_name: Optional[str] = None, column_name_format: str = "{name}[{category}]",
                 cat_missing_method: str = "fail", cat_missing_name: str = "(MISSING)",
                 _validated: bool = False):
        if cat_missing_method not in {"fail", "zero", "convert"}:
            raise ValueError(
                "cat_missing_method must be one of 'fail' 'zero' or 'convert'; "
                f" got {cat_missing_method}.")
        self._missing_method = cat_missing_method
        self._missing_category = cat_missing_name
        self._dev_codes = None
        if isinstance(cat_vec, torch.Tensor):
            # codes already in HBM: categories must be given (or a count)
            if categories is None:
                raise ValueError("categories are required when codes are a device tensor")
            self.categories = np.asarray(categories)
            self._dev_codes = D.to_dev(cat_vec, torch.int32)
            self._host_codes = None
            n = int(cat_vec.numel())
            self._has_missings = cat_missing_method == "zero"
            if n and not _validated:
                # the kernels index bins / vectors with the codes: check the range once (the
                # host path does the same, categorical_matrix.py:238-247)
                lo, hi = int(self._dev_codes.min().item()), int(self._dev_codes.max().item())
                if hi >= len(self.categories):
                    raise ValueError("Indices exceed length of categories.")
                if lo < -1:
                    raise ValueError("Indices must be non-negative (or -1 for missing).")
                if lo == -1 and cat_missing_method == "fail":
                    raise ValueError(
                        "Categorical data can't have missing values "
                        "if cat_missing_method='fail'.")
                if lo == -1 and cat_missing_method == "convert":
                    if cat_missing_name in self.categories:
                        raise ValueError(f"Missing category {cat_missing_name} already exists.")
                    self.categories = np.hstack([self.categories.astype(object),
                                                 [cat_missing_name]])
                    self._dev_codes = torch.where(self._dev_codes < 0,
                                                  len(self.categories) - 1, self._dev_codes).to(torch.int32)
                self._has_missings = lo == -1 and cat_missing_method == "zero"
        else:
            if hasattr(cat_vec, "cat") and hasattr(cat_vec.cat, "codes"):  # pandas categorical Series
                categories_ = np.asarray(cat_vec.cat.categories)
                indices = np.asarray(cat_vec.cat.codes)
                self.categories = categories_ if categories is None else np.asarray(categories)
            elif hasattr(cat_vec, "codes") and hasattr(cat_vec, "categories"):
                # a bare pd.Categorical: keep its declared categories and their order
                # (categorical_matrix.py:232-236 uses cat_vec.categories / .codes)
                categories_ = np.asarray(cat_vec.categories)
                indices = np.asarray(cat_vec.codes)
                self.categories = categories_ if categories is None else np.asarray(categories)
            elif categories is not None:
                self.categories = np.asarray(categories)
                indices = np.nan_to_num(np.asarray(cat_vec), nan=-1)
                if len(indices) and indices.max() >= len(self.categories):
                    raise ValueError("Indices exceed length of categories.")
                if len(indices) and indices.min() < -1:
                    raise ValueError("Indices must be non-negative (or -1 for missing).")
            else:
                indices, self.categories = _factorize(np.asarray(cat_vec))
            if np.any(indices == -1):
                if cat_missing_method == "fail":
                    raise ValueError(
                        "Categorical data can't have missing values "
                        "if cat_missing_method='fail'.")
                if cat_missing_method == "convert":
                    if cat_missing_name in self.categories:
                        raise ValueError(f"Missing category {cat_missing_name} already exists.")
                    self.categories = np.hstack([self.categories.astype(object),
                                                 [cat_missing_name]])
                    indices = np.where(indices < 0, len(self.categories) - 1, indices)
                    self._has_missings = False
                else:
                    self._has_missings = True
            else:
                self._has_missings = False
            try:
                self._host_codes = np.ascontiguousarray(indices).astype(np.int32, copy=False)
            except ValueError:
                raise ValueError(
                    "When creating a CategoricalMatrix with indices and categories, "
                    "indices must be castable to a numpy int32 dtype.")
            n = len(self._host_codes)
        self.drop_first = bool(drop_first)
        self.shape = (n, max(len(self.categories) - int(self.drop_first), 0))
        self.dtype = np.dtype(dtype)
        self._colname = column_name
        self._colname_format = column_name_format
        self._term = column_name if term_name is None else term_name

    __array_ufunc__ = None

    # ---- storage ------------------------------------------------------------------------
    @property
    def indices(self) -> np.ndarray:
        """int32 codes on the host (categorical_matrix.py:413-420)."""
        if self._host_codes is None:
            self._host_codes = D.to_host(self._dev_codes)
        return self._host_codes

    def _dev(self) -> torch.Tensor:
        if self._dev_codes is None:
            self._dev_codes = D.to_dev(self._host_codes, torch.int32)
        return self._dev_codes

    def to_device(self):
        """Upload the codes and take the ingest-time statistics now (the hot-level count that chooses the
        categorical x categorical kernel: a host synchronisation that must not happen inside a product -- a HIP
        graph capture, or a solver's timed loop)."""
        self._dev()
        self._hot_count()
        return self

    def _onehot(self, tdt=None):
        """(slab form of the one-hot encoding, row permutation) for the gather kernel; cached.
        tdt: dtype of the operand it will be multiplied with (the reference's cat x dense kernel
        is templated on d / mat_j only, ext/split.pyx:32-80); default the block's own dtype."""
        tdt = D.torch_dtype(self.dtype) if tdt is None else tdt
        cache = getattr(self, "_onehot_cache", None)
        if cache is None or cache[0] != tdt:   # astype() changes the nominal dtype in place
            from .ext._types import onehot_slab

            self._onehot_cache = (tdt, onehot_slab([(self._dev(), self.shape[1], self.drop_first)],
                                                   self.shape[0], tdt))
        return self._onehot_cache[1]

    def recover_orig(self):
        orig = self.categories[self.indices]
        if self._has_missings:
            orig = orig.view(np.ma.MaskedArray)
            orig.mask = self.indices == -1
        elif self._missing_method == "convert" and self._missing_category in self.categories:
            # categorical_matrix.py:463-468: the converted level reads back as missing
            orig = orig.view(np.ma.MaskedArray)
            orig.mask = self.indices == len(self.categories) - 1
        return orig

    def tocsr(self) -> sps.csr_matrix:
        """categorical_matrix.py:688-705."""
        col = self.indices.astype(np.int64) - int(self.drop_first)
        keep = col >= 0
        indptr = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
        return sps.csr_matrix((np.ones(int(keep.sum()), dtype=int), col[keep], indptr),
                              shape=self.shape)

    def to_sparse_matrix(self):
        return SparseMatrix(self.tocsr().astype(self.dtype))

    def toarray(self) -> np.ndarray:
        return self.tocsr().toarray()

    @property
    def cat(self):
        """The data as a pandas.Categorical (categorical_matrix.py:434-449; deprecated there, kept for callers that
        still unpack it)."""
        import warnings

        warnings.warn("This property will be removed in the next major release.", category=DeprecationWarning)
        try:
            import pandas as pd
        except ImportError as e:          # pragma: no cover
            raise ModuleNotFoundError("The `cat` property is provided for backward compatibility and "
                                      "requires pandas to be installed.") from e
        return pd.Categorical.from_codes(self.indices, categories=self.categories)

    def unpack(self):
        """categorical_matrix.py:719-721."""
        return self.cat

    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        """categorical_matrix.py:723-726: only the nominal dtype changes."""
        self.dtype = np.dtype(dtype)
        return self

    def getcol(self, i: int) -> SparseMatrix:
        i %= self.shape[1]
        col = (self.indices == i + int(self.drop_first)).astype(int)[:, None]
        return SparseMatrix(sps.csc_matrix(col))

    def __getitem__(self, item):
        row, col = check_indexer(item)
        full = (isinstance(col, slice) and len(range(*col.indices(self.shape[1]))) == self.shape[1]) or (
            isinstance(col, np.ndarray) and np.array_equal(col.ravel(), np.arange(self.shape[1])))
        if full and self._dev_codes is not None:
            # row indexing of codes that live in HBM stays in HBM
            kind, *arg = device_row_index(row, self.shape[0])
            codes = self._dev_codes[arg[0]:arg[1]] if kind == "slice" else \
                self._dev_codes[D.idx_dev(arg[0], torch.int64)]
            out = CategoricalMatrix(codes.contiguous(), categories=self.categories,
                                    drop_first=self.drop_first, dtype=self.dtype,
                                    column_name=self._colname, term_name=self._term,
                                    column_name_format=self._colname_format,
                                    cat_missing_method=self._missing_method,
                                    cat_missing_name=self._missing_category, _validated=True)
            # (the parent's hot-level count bounds the slice's when no row is repeated: no new count, no sync)
            if getattr(self, "_hot", None) is not None and (
                    kind == "slice" or np.unique(arg[0]).size == len(arg[0])):
                out._hot = min(self._hot, out.shape[0])
            return out
        if full:
            if isinstance(row, np.ndarray):
                row = row.ravel()
            return CategoricalMatrix(self.indices[row], categories=self.categories,
                                     drop_first=self.drop_first, dtype=self.dtype,
                                     column_name=self._colname,
                                     cat_missing_method=self._missing_method)
        return self.to_sparse_matrix()[row, col]

    def __repr__(self):
        return f"{self.__class__.__name__}\nCategories: {self.categories}"

    def get_names(self, type: str = "column", missing_prefix: Optional[str] = None, indices=None):
        name = self._colname if type == "column" else self._term
        k = self.shape[1]
        if name is None and missing_prefix is None:
            return [None] * k
        if name is None:
            idx = list(range(k)) if indices is None else indices
            name = f"{missing_prefix}{idx[0]}-{idx[-1]}"
        if type == "column":
            return [self._colname_format.format(name=name, category=c)
                    for c in self.categories[int(self.drop_first):]]
        return [name] * k

    def set_names(self, names, type: str = "column"):
        if isinstance(names, str):
            names = [names]
        if len(names) != 1 and not all(nm == names[0] for nm in names):
            raise ValueError("A categorical matrix has only one name")
        if type == "column":
            self._colname = names[0]
        else:
            self._term = names[0]

    def multiply(self, other) -> SparseMatrix:
        """Row scaling -> SparseMatrix (categorical_matrix.py:840-876)."""
        other = np.squeeze(np.asarray(other))
        if self.shape[0] != other.shape[0]:
            raise ValueError(
                f"Shapes do not match. Expected length of {self.shape[0]}. Got {len(other)}.")
        csr = self.tocsr().astype(other.dtype)
        return SparseMatrix(sps.diags(other) @ csr)

    # ---- hot path -----------------------------------------------------------------------
    def _matvec_dev(self, other, cols, out):
        """out[i] += other[col(i)] ; other / out device tensors of the same float dtype."""
        if cols is not None and D.nlen(cols) == self.shape[1]:
            cols = None
        if out is None:
            # fresh storage is written, not zero-filled and added to
            out = D.out_buf((self.shape[0],), other.dtype)
            xc.matvec_assign(self._dev(), other, self.shape[0], cols, self.shape[1], out, self.drop_first)
            return out
        xc.matvec(self._dev(), other, self.shape[0], cols, self.shape[1], out, self.drop_first)
        return out

    def matvec(self, other, cols=None, out=None):
        """out[i] += other[indices[i]] (categorical_matrix.py:495-541); 1-D `other` only."""
        check_matvec_out_shape(self, out)
        on_dev = D.is_dev(other)
        if not on_dev:
            other = np.asarray(other)
        if other.ndim > 1:
            raise NotImplementedError(
                "CategoricalMatrix.matvec is only implemented for 1d arrays.")
        check_matvec_dimensions(self, other, transpose=False)
        cols_n = normalize_index(cols, self.shape[1])
        is_int = (not on_dev) and np.issubdtype(other.dtype, np.signedinteger)
        if on_dev:
            fdt = other.dtype if other.dtype in (torch.float32, torch.float64) else torch.float64
        else:
            fdt = D.torch_dtype(other.dtype) if other.dtype in (np.float32, np.float64) else torch.float64
        res = self._matvec_dev(D.to_dev(other, fdt), D.idx_dev(cols_n), None)
        if on_dev:
            if out is None:
                return res
            out += res
            return out
        res = D.to_host(res)
        if out is not None:
            out += res
            res = out
        return res.astype(int) if is_int else res

    def _det_plan(self):
        """(perm, bstart, n_blocks, cat_bptr) of the deterministic transpose_matvec: rows grouped
        by column once (stable device sort at ingest), runs cut into fixed blocks."""
        if getattr(self, "_det", None) is None:
            from ._lib import lib

            blk = int(lib().tm_cat_det_block_rows())
            codes = self._dev().to(torch.int64) - int(self.drop_first)
            k = self.shape[1]
            valid = (codes >= 0) & (codes < k)
            rows = torch.nonzero(valid, as_tuple=False).flatten()
            order = torch.sort(codes[rows], stable=True).indices
            perm = rows[order].to(torch.int32).contiguous()
            cnt = torch.bincount(codes[rows], minlength=k) if rows.numel() else \
                torch.zeros(k, dtype=torch.int64, device=codes.device)
            nb = torch.div(cnt + blk - 1, blk, rounding_mode="floor")
            cat_bptr = torch.zeros(k + 1, dtype=torch.int64, device=codes.device)
            torch.cumsum(nb, dim=0, out=cat_bptr[1:])
            n_blocks = int(cat_bptr[-1].item())
            seg0 = torch.cumsum(cnt, dim=0) - cnt                       # first element of a column
            col_of_blk = torch.repeat_interleave(torch.arange(k, device=codes.device), nb)
            within = torch.arange(n_blocks, device=codes.device) - cat_bptr[:-1][col_of_blk]
            bstart = torch.empty(n_blocks + 1, dtype=torch.int64, device=codes.device)
            bstart[:-1] = seg0[col_of_blk] + within * blk
            bstart[-1] = int(cnt.sum().item())
            self._det = (perm, bstart.contiguous(), n_blocks, cat_bptr.contiguous())
        return self._det

    def _transpose_matvec_dev(self, vec, rows, cols, out_full):
        """out_full (length n_cols, device) += ...; returns out_full."""
        if DETERMINISTIC and self.shape[0] > 0:
            # bitwise reproducible sums (the reference's K4a is deterministic); a row restriction
            # becomes a masked vector, a column restriction a masked result
            vec = D.masked_d(vec, rows)        # (a repeated row counts per occurrence, as the reference's loop does)
            perm, bstart, n_blocks, cat_bptr = self._det_plan()
            tgt = out_full if cols is None else D.zeros((self.shape[1],), out_full.dtype)
            xc.transpose_matvec_det(perm, bstart, n_blocks, cat_bptr, self.shape[1], vec, tgt,
                                    accumulate=cols is None)
            if cols is not None:
                c64 = cols.to(torch.int64)
                out_full[c64] += tgt[c64]
            return out_full
        xc.transpose_matvec(self._dev(), vec, self.shape[1], rows, cols, out_full, self.drop_first)
        return out_full

    def transpose_matvec(self, vec, rows=None, cols=None, out=None):
        """categorical_matrix.py:543-616."""
        on_dev = D.is_dev(vec)
        if not on_dev:
            vec = np.asarray(vec)
        check_matvec_dimensions(self, vec, transpose=True)
        if vec.ndim > 1:
            raise NotImplementedError(
                "CategoricalMatrix.transpose_matvec is only implemented for 1d arrays.")
        check_transpose_matvec_out_shape(self, out)
        rows_n = normalize_index(rows, self.shape[0])
        cols_n = normalize_index(cols, self.shape[1])
        if on_dev:
            fdt = vec.dtype if vec.dtype in (torch.float32, torch.float64) else torch.float64
        else:
            fdt = D.torch_dtype(vec.dtype) if vec.dtype in (np.float32, np.float64) else D.torch_dtype(self.dtype)
        full = D.zeros((self.shape[1],), fdt)
        if (rows_n is None or len(rows_n) > 0) and (cols_n is None or len(cols_n) > 0):
            self._transpose_matvec_dev(D.to_dev(vec, fdt), D.idx_dev(rows_n), D.idx_dev(cols_n), full)
        if out is not None:
            out += full if D.is_dev(out) else D.to_host(full).astype(out.dtype, copy=False)
            return out
        if cols_n is not None:
            full = full[D.idx_dev(cols_n, torch.int64)]
        return full if on_dev else D.to_host(full)

    def _sandwich_diag_dev(self, d, rows, cols):
        """Diagonal of X' diag(d) X as a device vector (restricted to cols)."""
        if DETERMINISTIC and self.shape[0] > 0 and self.shape[1] > 0:
            diag = D.zeros((self.shape[1],), d.dtype)
            self._transpose_matvec_dev(d, rows, None, diag)
        else:
            diag = xc.sandwich_categorical(self._dev(), d, rows, self.shape[1], self.drop_first)
        if cols is not None and D.nlen(cols) < self.shape[1]:
            diag = diag[cols.to(torch.int64)]
        return diag

    def sandwich(self, d, rows=None, cols=None) -> sps.dia_matrix:
        """Diagonal sandwich, returned as scipy dia_matrix like the reference
        (categorical_matrix.py:618-653)."""
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        diag = self._sandwich_diag_dev(D.to_dev(d), D.idx_dev(normalize_index(rows, self.shape[0])),
                                       D.idx_dev(normalize_index(cols, self.shape[1])))
        return sps.diags(D.to_host(diag))

    def _cross_sandwich_dev(self, other, d, rows, L_cols, R_cols):
        """X' diag(d) Y, device in / device out (categorical_matrix.py:655-671)."""
        if isinstance(other, DenseMatrix):
            if self.shape[0] >= 4096 and self.shape[1] > 0:
                # large n: one pass over the whole dense block (masked d for a row restriction,
                # sub-selection of the small result for column restrictions) -- the wide-load LDS
                # tile kernel when the levels fit one tile, else the atomic-free gather kernel on
                # the one-hot slab
                from .ext import sparse as xs

                d = D.masked_d(d, rows)
                cats = [(self._dev(), self.shape[1], self.drop_first)]
                if (other.dtype == self.dtype and d.dtype == other._dev_c().buf.dtype
                        and xsplit.multi_cat_dense_wide_ok(cats, other._dev_c())):
                    res = xsplit.multi_cat_dense_sandwich(cats, d, other._dev_c())
                elif (other.dtype == self.dtype and d.dtype == other._dev_c().buf.dtype
                      and xsplit.multi_cat_dense_tile_ok(cats, other._dev_c())):
                    res = xsplit.multi_cat_dense_sandwich(cats, d, other._dev_c())    # narrow operand
                elif (d.dtype == other._dev_c().buf.dtype
                      and xsplit.cat_dense_sorted_ok(other._dev_c())):
                    # more levels than one LDS tile holds: rows grouped by level, one pass over
                    # the dense block whatever the number of levels
                    res = xsplit.cat_dense_sandwich_sorted(self._det_plan(), self.shape[1], d,
                                                           other._dev_c())
                else:
                    oh, inv = self._onehot(other._dev_c().buf.dtype)
                    res = xs.csr_dense_sandwich_slab(oh, other._dev_c(), d)[inv]
                return self._restrict(res, L_cols, R_cols)
            res = xsplit.sandwich_cat_dense(self._dev(), self.shape[1], d, other._dev(), rows,
                                            R_cols, self.drop_first)
            return self._restrict(res, L_cols, None)
        if isinstance(other, SparseMatrix):
            S = other._dev()
            n_out = other.shape[1] if R_cols is None else D.nlen(R_cols)
            if (self.shape[0] >= 4096 and self.shape[1] * n_out * 8 > 128 * 1024
                    and d.dtype == S.data.dtype and S.data.numel() > 0):
                # the [levels][columns] tile would take several passes over the rows: level-sorted
                # kernel instead (masked d for a row restriction, sub-selection of the result)
                d = D.masked_d(d, rows)
                res = xsplit.cat_sparse_sandwich_sorted(self._det_plan(), self.shape[1], d, S)
                return self._restrict(res, L_cols, R_cols)
            res = xsplit.sandwich_cat_sparse(self._dev(), self.shape[1], d, S, rows,
                                             R_cols, self.drop_first)
            return self._restrict(res, L_cols, None)
        if isinstance(other, CategoricalMatrix):
            if (other is not self and d.dtype in (torch.float32, torch.float64)
                    and xsplit.cat_cat_sorted_pays(self.shape[0], self.shape[1], other.shape[1])):
                # a table of several LDS tiles: rows grouped by this block's level (static twin of the pair), every
                # tile reads only its own rows; a row restriction is a masked d
                d = D.masked_d(d, rows)
                res = xsplit.sandwich_cat_cat_sorted(self._sorted_pair(other), self.shape[1], other.shape[1], d)
                return self._restrict(res, L_cols, R_cols)
            res = xsplit.sandwich_cat_cat(self._dev(), other._dev(), self.shape[1],
                                          other.shape[1], d, rows, self.drop_first,
                                          other.drop_first, hot=min(self._hot_count(), other._hot_count()))
            return self._restrict(res, L_cols, R_cols)
        raise TypeError

    def _sorted_pair(self, other):
        """(ci_sorted, cj_sorted, perm, lptr) for tm_cat_cat_sandwich_sorted_*: this block's rows in level order
        (the perm of _det_plan), both blocks' column indices in that order, the first position of every level.
        Static per pair; cached by the partner's identity."""
        cache = self.__dict__.setdefault("_sorted_pairs", {})
        key = id(other)
        hit = cache.get(key)
        if hit is None or hit[0] is not other:
            perm = self._det_plan()[0]
            p64 = perm.to(torch.int64)
            ci_s = (self._dev()[p64] - int(self.drop_first)).to(torch.int32).contiguous()
            cj = other._dev()[p64].to(torch.int32) - int(other.drop_first)
            cj_s = torch.where((cj >= 0) & (cj < other.shape[1]), cj, torch.full_like(cj, -1)).contiguous()
            cnt = torch.bincount(ci_s.to(torch.int64), minlength=self.shape[1]) if ci_s.numel() else \
                torch.zeros(self.shape[1], dtype=torch.int64, device=ci_s.device)
            lptr = torch.zeros(self.shape[1] + 1, dtype=torch.int64, device=ci_s.device)
            torch.cumsum(cnt, dim=0, out=lptr[1:])
            hit = cache[key] = (other, (ci_s, cj_s, perm, lptr.contiguous()))
        return hit[1]

    def _hot_count(self) -> int:
        """Rows of the most frequent level (cached): an upper bound of what one cell of a
        categorical x categorical table collects."""
        h = getattr(self, "_hot", None)
        if h is None:
            c = self._dev()
            h = self._hot = int(torch.bincount(c.to(torch.int64).clamp_(min=0)).max().item()) if c.numel() else 0
        return h

    @staticmethod
    def _restrict(res, L_cols, R_cols):
        """categorical_matrix.py:296-316 (_row_col_indexing) on the small device result."""
        if L_cols is not None and D.nlen(L_cols) != res.shape[0]:
            res = res[L_cols.to(torch.int64)]
        if R_cols is not None and D.nlen(R_cols) != res.shape[1]:
            res = res[:, R_cols.to(torch.int64)]
        return res

    def _cross_sandwich(self, other, d, rows=None, L_cols=None, R_cols=None):
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        res = self._cross_sandwich_dev(
            other, D.to_dev(d), D.idx_dev(normalize_index(rows, self.shape[0])),
            D.idx_dev(normalize_index(L_cols, self.shape[1])),
            D.idx_dev(normalize_index(R_cols, other.shape[1])))
        return res if on_dev else D.to_host(res)

    def _cross_dense(self, other, d, rows, L_cols, R_cols):
        return self._cross_sandwich(DenseMatrix(other), d, rows, L_cols, R_cols)

    def _cross_categorical(self, other, d, rows, L_cols, R_cols):
        if not isinstance(other, CategoricalMatrix):
            raise TypeError
        return self._cross_sandwich(other, d, rows, L_cols, R_cols)

    def _get_col_stds(self, weights, col_means):
        """categorical_matrix.py:728-737: X_ij in {0,1}, so E[X^2] = E[X]."""
        mean = self.transpose_matvec(weights)
        return np.sqrt(np.maximum(mean - col_means**2, 0))
