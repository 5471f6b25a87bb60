"""Ingest: build device-backed blocks from a CSC matrix or a pandas DataFrame.

Mirrors the reference's `from_csc` / `from_df` / `from_pandas` (src/tabmat/constructor.py:29-212,
297-308; constructor_util.py:11-49) — same parameters, thresholds, block order, column indices and
names — so that `tabmat_amd.from_pandas(df)` is a drop-in for the object a GLM solver is handed.
`from_df` is one-off host work on a pandas frame; its blocks upload themselves to
HBM on first use (`to_device()` forces it).  `from_csc` also takes storage that already lives in HBM
(a device-resident SparseMatrix or raw CSC arrays on the device) and splits it there.
`from_formula` needs the third-party `formulaic` package and is out of
scope."""
from __future__ import annotations

import warnings
from typing import Optional, Sequence

import numpy as np
from scipy import sparse as sps

from .categorical_matrix import CategoricalMatrix
from .dense_matrix import DenseMatrix
from .matrix_base import MatrixBase
from .sparse_matrix import SparseMatrix
from .split_matrix import SplitMatrix


def _split_sparse_and_dense_parts(arg1: sps.csc_matrix, threshold: float = 0.1,
                                  column_names: Optional[Sequence] = None,
                                  term_names: Optional[Sequence] = None):
    """Columns denser than `threshold` -> one F-ordered DenseMatrix, the rest -> one SparseMatrix;
    also returns the column positions of both parts (constructor_util.py:11-49)."""
    if not isinstance(arg1, sps.csc_matrix):
        raise TypeError("X must be of type scipy.sparse.csc_matrix or matrix.SparseMatrix,"
                        f"not {type(arg1)}")
    if not 0 <= threshold <= 1:
        raise ValueError("Threshold must be between 0 and 1.")
    n = arg1.shape[0]
    density = np.diff(arg1.indptr) / n
    is_dense = density > threshold
    dense_idx = np.flatnonzero(is_dense)
    sparse_idx = np.flatnonzero(~is_dense)
    cn = [None] * arg1.shape[1] if column_names is None else list(column_names)
    tn = cn if term_names is None else list(term_names)
    dense = DenseMatrix(np.asfortranarray(arg1[:, dense_idx].toarray()),
                        column_names=[cn[i] for i in dense_idx],
                        term_names=[tn[i] for i in dense_idx])
    sparse = SparseMatrix(arg1[:, sparse_idx], column_names=[cn[i] for i in sparse_idx],
                          term_names=[tn[i] for i in sparse_idx])
    return dense, sparse, dense_idx, sparse_idx


def _split_device_csr(csr, threshold: float, column_names=None, term_names=None):
    """The same split for a block whose CSR twin already lives in HBM (CsrDev): the dense part
    is written out as a row-major device array, the sparse part stays a CSR twin -- only the
    per-column counts (m integers) visit the host, never the entries (SURVEY.md 8f-3;
    constructor_util.py:11-49)."""
    import torch

    from .ext._types import CsrDev

    if not 0 <= threshold <= 1:
        raise ValueError("Threshold must be between 0 and 1.")
    n, m = csr.n, csr.m
    dev = csr.data.device
    nnz = int(csr.data.numel())
    cols64 = csr.indices.to(torch.int64)
    col_cnt = torch.bincount(cols64, minlength=m) if nnz else torch.zeros(m, dtype=torch.int64, device=dev)
    is_dense = (col_cnt.to(torch.float64) / max(n, 1)) > threshold
    dense_idx = torch.nonzero(is_dense).ravel()
    sparse_idx = torch.nonzero(~is_dense).ravel()
    kd, ks = int(dense_idx.numel()), int(sparse_idx.numel())
    new_id = torch.empty(m, dtype=torch.int64, device=dev)
    new_id[dense_idx] = torch.arange(kd, device=dev)
    new_id[sparse_idx] = torch.arange(ks, device=dev)
    row_cnt = csr.indptr[1:] - csr.indptr[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), row_cnt)
    to_dense = is_dense[cols64]
    T = torch.zeros((n, kd), dtype=csr.data.dtype, device=dev)
    if kd and nnz:
        T[rows[to_dense], new_id[cols64[to_dense]]] = csr.data[to_dense]
    keep = ~to_dense
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if nnz:
        torch.cumsum(torch.bincount(rows[keep], minlength=n), dim=0, out=indptr[1:])
    # (the column map is monotone on the kept columns: rows stay in canonical order)
    sp = CsrDev(csr.data[keep].contiguous(), new_id[cols64[keep]].to(torch.int32).contiguous(), indptr, n, ks)
    di, si = dense_idx.cpu().numpy(), sparse_idx.cpu().numpy()
    cn = [None] * m if column_names is None else list(column_names)
    tn = cn if term_names is None else list(term_names)
    dense = DenseMatrix(T, column_names=[cn[i] for i in di], term_names=[tn[i] for i in di])
    sparse = SparseMatrix.from_device(sp)
    sparse._init_names([cn[i] for i in si], [tn[i] for i in si])
    return dense, sparse, di, si


def csc_arrays_to_csr_dev(data, indices, indptr, shape):
    """CSC arrays (torch cuda tensors or numpy: values, row indices, column pointers) -> CsrDev in
    HBM.  The entries must be canonical (rows ascending inside a column, no duplicates)."""
    import torch

    from . import _device as D
    from .ext._types import CsrDev

    n, m = int(shape[0]), int(shape[1])
    data = D.to_dev(data)
    rows = D.to_dev(indices).to(torch.int64)
    ptr = D.to_dev(indptr).to(torch.int64)
    dev = data.device
    cols = torch.repeat_interleave(torch.arange(m, device=dev, dtype=torch.int32), ptr[1:] - ptr[:-1])
    order = torch.sort(rows, stable=True).indices          # columns stay ascending inside a row
    out_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if rows.numel():
        torch.cumsum(torch.bincount(rows, minlength=n), dim=0, out=out_ptr[1:])
    return CsrDev(data[order].contiguous(), cols[order].contiguous(), out_ptr, n, m)


def from_csc(mat, threshold=0.1, column_names=None, term_names=None):
    """CSC matrix -> SplitMatrix([dense columns, sparse columns]) (constructor.py:297-308).
    `mat`: a scipy.sparse.csc_matrix (host ingest, blocks upload on first use), a SparseMatrix
    whose storage already lives in HBM, or a tuple (data, indices, indptr, shape) of CSC arrays on
    the device -- the last two are split ON the device, no entry visits the host."""
    if isinstance(mat, tuple) and len(mat) == 4:
        csr = csc_arrays_to_csr_dev(*mat)
        dense, sparse, dense_idx, sparse_idx = _split_device_csr(csr, threshold, column_names, term_names)
        return SplitMatrix([dense, sparse], [dense_idx, sparse_idx])
    if isinstance(mat, SparseMatrix):
        if mat._array is None and mat._devblk is not None:
            dense, sparse, dense_idx, sparse_idx = _split_device_csr(
                mat._dev(), threshold, column_names or mat._colnames, term_names or mat._terms)
            return SplitMatrix([dense, sparse], [dense_idx, sparse_idx])
        mat = mat._host()
    dense, sparse, dense_idx, sparse_idx = _split_sparse_and_dense_parts(mat, threshold, column_names,
                                                                         term_names)
    return SplitMatrix([dense, sparse], [dense_idx, sparse_idx])


def _is_categorical(col) -> bool:
    return str(col.dtype) == "category"


def _is_stringlike(col) -> bool:
    import pandas as pd

    return col.dtype == object or isinstance(col.dtype, pd.StringDtype)


def from_df(df, dtype=np.float64, sparse_threshold: float = 0.1, cat_threshold: int = 4,
            object_as_cat: bool = False, cat_position: str = "expand", drop_first: bool = False,
            categorical_format: str = "{name}[{category}]", cat_missing_method: str = "fail",
            cat_missing_name: str = "(MISSING)") -> MatrixBase:
    """pandas DataFrame -> SplitMatrix (constructor.py:29-212): categorical columns become
    CategoricalMatrix blocks (one-hot dense/sparse columns below `cat_threshold` levels), numeric
    and boolean columns are collected into one dense and one sparse block by their share of
    nonzeros, anything else is ignored with a warning.  `cat_position='end'` moves all
    categorical columns behind the others."""
    import pandas as pd

    if cat_position not in ("expand", "end"):
        raise ValueError("cat_position must be 'expand' or 'end'")
    blocks, positions, from_cat = [], [], []
    dense_cols, dense_pos, sparse_cols, sparse_pos, ignored = [], [], [], [], []
    nxt = 0            # next column position in the assembled matrix ('expand' ordering)

    for j, name in enumerate(df.columns):
        col = df.iloc[:, j]
        if object_as_cat and _is_stringlike(col):
            col = col.astype("category")
        if isinstance(col.dtype, pd.SparseDtype):
            sparse_cols.append(j)
            sparse_pos.append(nxt)
            nxt += 1
        elif _is_categorical(col):
            cat = CategoricalMatrix(col, drop_first=drop_first, dtype=dtype, column_name=name,
                                    term_name=name, column_name_format=categorical_format,
                                    cat_missing_method=cat_missing_method,
                                    cat_missing_name=cat_missing_name)
            if len(cat.categories) < cat_threshold:
                dense, sparse, di, si = _split_sparse_and_dense_parts(
                    sps.csc_matrix(cat.tocsr(), dtype=dtype), threshold=sparse_threshold,
                    column_names=cat.get_names("column"), term_names=cat.get_names("term"))
                new = [(dense, di), (sparse, si)]
            else:
                new = [(cat, np.arange(cat.shape[1]))]
            for blk, local in new:
                blocks.append(blk)
                from_cat.append(True)
                positions.append(nxt + local if cat_position == "expand" else local)
            if cat_position == "expand":
                nxt += sum(len(local) for _, local in new)
        elif col.dtype == bool or pd.api.types.is_numeric_dtype(col.dtype):
            share = float((col != (False if col.dtype == bool else 0)).mean()) if len(col) else 0.0
            if share <= sparse_threshold:
                sparse_cols.append(j)
                sparse_pos.append(nxt)
            else:
                dense_cols.append(j)
                dense_pos.append(nxt)
            nxt += 1
        else:
            ignored.append(name)

    if ignored:
        warnings.warn(f"Columns {ignored} were ignored. Make sure they have a valid dtype.")
    names = np.asarray(df.columns)
    if dense_cols:
        blocks.append(DenseMatrix(df.iloc[:, dense_cols].to_numpy().astype(dtype, copy=False),
                                  column_names=names[dense_cols], term_names=names[dense_cols]))
        positions.append(np.asarray(dense_pos))
        from_cat.append(False)
    if sparse_cols:
        blocks.append(SparseMatrix(sps.coo_matrix(df.iloc[:, sparse_cols], dtype=dtype), dtype=dtype,
                                   column_names=names[sparse_cols], term_names=names[sparse_cols]))
        positions.append(np.asarray(sparse_pos))
        from_cat.append(False)

    if cat_position == "end":          # categorical-derived blocks go behind the nxt plain columns
        shifted = []
        for pos, is_cat in zip(positions, from_cat):
            if is_cat:
                shifted.append(np.asarray(pos) + nxt)
                nxt += len(pos)
            else:
                shifted.append(pos)
        positions = shifted

    if not blocks:
        raise ValueError("DataFrame contained no valid column")
    return SplitMatrix(blocks, positions) if len(blocks) > 1 else blocks[0]


def from_pandas(df, dtype=np.float64, sparse_threshold: float = 0.1, cat_threshold: int = 4,
                object_as_cat: bool = False, cat_position: str = "expand", drop_first: bool = False,
                categorical_format: str = "{name}[{category}]", cat_missing_method: str = "fail",
                cat_missing_name: str = "(MISSING)") -> MatrixBase:
    """Deprecated alias of `from_df` in the reference (constructor.py:215-283)."""
    return from_df(df, dtype=dtype, sparse_threshold=sparse_threshold, cat_threshold=cat_threshold,
                   object_as_cat=object_as_cat, cat_position=cat_position, drop_first=drop_first,
                   categorical_format=categorical_format, cat_missing_method=cat_missing_method,
                   cat_missing_name=cat_missing_name)
