"""Helpers for the -m gpu parity tests: build tabmat_amd blocks from the neutral specs of
tests/_cases.py and compare against the CPU oracle."""
import numpy as np

import _cases as cs


def to_tm_block(spec, dtype=None):
    import tabmat_amd as tm

    kind = spec[0]
    if kind == "dense":
        X = spec[1] if dtype is None else spec[1].astype(dtype)
        return tm.DenseMatrix(X)
    if kind == "sparse":
        S = spec[1] if dtype is None else spec[1].astype(dtype)
        return tm.SparseMatrix(S)
    codes, ncat, drop = spec[1], spec[2], spec[3]
    missing = bool((codes < 0).any())
    return tm.CategoricalMatrix(codes, categories=np.arange(ncat), drop_first=drop,
                                dtype=dtype or np.float64,
                                cat_missing_method="zero" if missing else "fail")


def to_tm_split(specs, idx, dtype=None):
    import tabmat_amd as tm

    return tm.SplitMatrix([to_tm_block(s, dtype) for s in specs], [np.asarray(i) for i in idx])


def sub(A, rows, cols):
    if rows is not None:
        A = A[np.asarray(rows, dtype=int), :]
    if cols is not None:
        A = A[:, np.asarray(cols, dtype=int)]
    return A


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    return float(np.abs(a - b).max() / den) if b.size else 0.0


def nat_err(a, b):
    """Entry-wise error of a sandwich at its NATURAL scale: max_ij |a_ij - b_ij| / sqrt(b_ii b_jj) (for a
    nonnegative weight vector |S_ij| <= sqrt(S_ii S_jj), so this is the error relative to the largest value the
    entry could have taken).  A wrong small block -- a categorical x categorical cell, a sparse x categorical
    strip -- cannot hide under the magnitude of the dense block the way it can under max|b| (rel_err); the
    reference's own tests compare entry by entry (tests/test_split_matrix.py:170-189).  An entry whose row or
    column has a zero diagonal must be reproduced exactly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape and a.ndim == 2 and a.shape[0] == a.shape[1]
    if not b.size:
        return 0.0
    dg = np.sqrt(np.abs(np.diag(b)))
    den = np.outer(dg, dg)
    err = np.abs(a - b)
    out = np.where(den > 0, err / np.where(den > 0, den, 1.0), np.where(err == 0, 0.0, np.inf))
    return float(out.max())
