"""Ingest: build device-backed blocks from a CSC matrix or a pandas DataFrame.

Mirrors the reference's `from_csc` / `from_df` / `from_pandas` (src/tabmat/constructor.py:29-212,
297-308; constructor_util.py:11-49) — same parameters, thresholds, block order, column indices and
names — so that `tabmat_amd.from_pandas(df)` is a drop-in for the object a GLM solver is handed.
This is one-off host work; the blocks upload themselves to HBM on first use
(`to_device()` forces it).  `from_formula` needs the third-party `formulaic` package and is out of
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


def from_csc(mat: sps.csc_matrix, threshold=0.1, column_names=None, term_names=None):
    """CSC matrix -> SplitMatrix([dense columns, sparse columns]) (constructor.py:297-308)."""
    dense, sparse, dense_idx, sparse_idx = _split_sparse_and_dense_parts(mat, threshold)
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
