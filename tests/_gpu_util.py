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


def wdiag(M, d, rows=None, cols=None):
    """diag(M[rows, cols]' diag(d[rows]) M[rows, cols]) in float64 for a host operand M: a dense array, a
    scipy.sparse matrix, or a categorical spec ("cat", codes, n_columns, drop_first) (one-hot, -1 = missing)."""
    from scipy import sparse as sps

    d = np.asarray(d, dtype=np.float64)
    r = slice(None) if rows is None else np.asarray(rows, dtype=np.int64)
    if isinstance(M, tuple):
        _, codes, ncol, drop = M
        code = np.asarray(codes)[r].astype(np.int64) - (1 if drop else 0)
        ok = code >= 0
        out = np.bincount(code[ok], weights=d[r][ok], minlength=ncol)[:ncol]
    elif sps.issparse(M):
        Mr = M.tocsr()[r].astype(np.float64)
        out = np.asarray(Mr.multiply(Mr).T @ d[r]).ravel()
    else:
        Mr = np.asarray(M, dtype=np.float64)[r]
        out = (Mr * Mr).T @ d[r]
    return out if cols is None else out[np.asarray(cols, dtype=np.int64)]


def cross_err(a, b, d, L, R, rows=None, lc=None, rc=None):
    """Entry-wise error of a CROSS sandwich L' D R at its natural scale sqrt((L'DL)_ii (R'DR)_jj) (Cauchy-Schwarz
    bound of the entry for nonnegative d): the rectangular counterpart of nat_err -- a wrong strip cannot hide
    under the largest entry of the block.  L / R: the host operands (see wdiag)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not b.size:
        return 0.0
    den = np.sqrt(np.outer(np.abs(wdiag(L, d, rows, lc)), np.abs(wdiag(R, d, rows, rc))))
    err = np.abs(a - b)
    out = np.where(den > 0, err / np.where(den > 0, den, 1.0), np.where(err == 0, 0.0, np.inf))
    return float(out.max())
