"""Argument normalisation and the error conventions of the reference's util.py
(/root/reference/src/tabmat/util.py:6-67): which exception type is raised for which
mismatch, and the message fragments the reference's tests match on."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

_T2NP = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
         torch.int32: np.dtype(np.int32), torch.int64: np.dtype(np.int64)}


def np_dtype_of(x) -> np.dtype:
    """numpy dtype of a numpy array or torch tensor."""
    if isinstance(x, torch.Tensor):
        return _T2NP.get(x.dtype, np.dtype(object))
    return np.asarray(x).dtype


def set_up_rows_or_cols(arr, length: int, dtype=np.int32) -> np.ndarray:
    """None -> arange(length); anything else -> integer array (util.py:6-12)."""
    return np.arange(length, dtype=dtype) if arr is None else np.asarray(arr).astype(dtype)


def setup_restrictions(shape, rows, cols, dtype=np.int32):
    """util.py:15-24."""
    return (set_up_rows_or_cols(rows, shape[0], dtype), set_up_rows_or_cols(cols, shape[1], dtype))


def normalize_index(arr, length: int) -> Optional[np.ndarray]:
    """Device-side convention: None means "all" and stays None (no O(n) arange);
    lists/arrays become int32.  `length` is only used for bounds checking."""
    if arr is None:
        return None
    out = np.asarray(arr).astype(np.int32).reshape(-1)
    if out.size and (out.min() < -length or out.max() >= length):
        raise IndexError("index out of range")
    if out.size and out.min() < 0:
        out = np.where(out < 0, out + length, out).astype(np.int32)
    return out


def collapse_identity(idx: Optional[np.ndarray], length: int) -> Optional[np.ndarray]:
    """A column selection that lists every column in order (glum passes its active set as
    `cols=np.arange(p)` when nothing is excluded) is the unrestricted product: None, so that the
    call takes the unrestricted kernels.  O(len) on the host, meant for COLUMN indices."""
    if idx is not None and idx.size == length and length > 0 and idx[0] == 0 \
            and idx[-1] == length - 1 and bool((idx[1:] > idx[:-1]).all()):
        return None
    return idx


def _first_dim_mismatch(out, expected: int) -> None:
    if out is not None and out.shape[0] != expected:
        raise ValueError(
            f"The first dimension of 'out' must be {expected}, but it is {out.shape[0]}."
        )


def check_transpose_matvec_out_shape(mat, out) -> None:
    """util.py:35-37."""
    _first_dim_mismatch(out, mat.shape[1])


def check_matvec_out_shape(mat, out) -> None:
    """util.py:40-42."""
    _first_dim_mismatch(out, mat.shape[0])


def check_matvec_dimensions(mat, vec, transpose: bool) -> None:
    """util.py:45-52."""
    dim = 0 if transpose else 1
    if mat.shape[dim] != vec.shape[0]:
        raise ValueError(
            f"shapes {tuple(mat.shape)} and {tuple(vec.shape)} not aligned: "
            f"{mat.shape[dim]} (dim {dim}) != {vec.shape[0]} (dim 0)"
        )


def check_sandwich_compatible(mat, d) -> None:
    """util.py:55-67: ValueError on length mismatch, TypeError when dtypes differ."""
    if mat.shape[0] != d.shape[0]:
        raise ValueError(
            f"shapes {tuple(mat.shape)} and {tuple(d.shape)} not aligned: "
            f"{mat.shape[0]} (dim 0) != {d.shape[0]} (dim 0)"
        )
    dt = np_dtype_of(d)
    if np.dtype(mat.dtype) != dt:
        raise TypeError(
            "self and d need to be of same dtype, either np.float64 or np.float32. "
            f"self is of type {mat.dtype}, while d is of type {dt}."
        )


def selects_all_columns(col, m: int) -> bool:
    """True when the canonical column indexer keeps every column in order."""
    if isinstance(col, slice):
        return col.indices(m) == (0, m, 1) or (m == 0)
    col = np.asarray(col).ravel()
    return len(col) == m and bool(np.array_equal(col, np.arange(m)))


def device_row_index(row, n: int):
    """Canonical row indexer -> ("slice", lo, hi) for a unit-step slice, else ("index", int64
    numpy array of non-negative row ids).  Used by the device form of row indexing
    (split_matrix.py:462-477: train / validation splits without a host round trip)."""
    if isinstance(row, slice):
        lo, hi, step = row.indices(n)
        if step == 1:
            return ("slice", lo, max(lo, hi))
        return ("index", np.arange(lo, hi, step, dtype=np.int64))
    r = np.asarray(row).ravel()
    if r.dtype == bool:
        r = np.flatnonzero(r)
    r = r.astype(np.int64)
    r = np.where(r < 0, r + n, r)
    if r.size and (int(r.min()) < 0 or int(r.max()) >= n):
        # the host path raises from numpy; on the device this would be a silent out-of-bounds gather
        bad = int(r[(r < 0) | (r >= n)][0])
        raise IndexError(f"index {bad if bad < n else bad} is out of bounds for axis 0 with size {n}")
    return ("index", r)


def check_indexer(indexer):
    """Canonical (row_indexer, col_indexer) pair (util.py:70-115)."""
    if not isinstance(indexer, tuple):
        indexer = (indexer, slice(None))
    if len(indexer) > 2:
        raise ValueError("More than two indexers are not supported.")
    row, col = indexer
    r_slice, c_slice = isinstance(row, slice), isinstance(col, slice)
    if r_slice and c_slice:
        return row, col
    if r_slice:
        col = np.asarray(col)
        if col.ndim > 1:
            raise ValueError("Indexing would result in a matrix with more than 2 dimensions.")
        return row, col.reshape(-1)
    if c_slice:
        row = np.asarray(row)
        if row.ndim > 1:
            raise ValueError("Indexing would result in a matrix with more than 2 dimensions.")
        return row.reshape(-1), col
    row, col = np.asarray(row), np.asarray(col)
    if row.ndim <= 1 and col.ndim <= 1:
        return np.ix_(row.reshape(-1), col.reshape(-1))
    if row.ndim == 2 and row.shape[1] == 1 and col.ndim == 2 and col.shape[0] == 1:
        return row, col
    raise ValueError("This type of indexing is not supported.")
