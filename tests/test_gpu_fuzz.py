"""-m gpu: seeded random SplitMatrix configurations (block mix, shapes, storage orders, dtypes,
drop_first / missing codes, row and column restrictions) against a dense float64 numpy evaluation
of the same matrix -- a net under the specialised fast paths (streams, twins, packed tiles), whose
dispatch depends on shapes and alignments.  Tolerances: float64 1e-10 relative (north_star), float32
blocks 5e-4 relative to the largest entry of the result."""
import os

import numpy as np
import pytest
from scipy import sparse as sps

pytestmark = pytest.mark.gpu


def _random_split(rng, dtype):
    import tabmat_amd as tm

    n = int(rng.choice([1, 7, 64, 129, 1000, 4096, 5003, 20000]))
    blocks, dense_parts = [], []
    kinds = list(rng.permutation(["dense", "sparse", "cat", "cat", "sparse", "dense"])[: rng.integers(1, 6)])
    if rng.random() < 0.3:          # a categorical-heavy design (fused pair tables, fused matvec)
        kinds += ["cat_small"] * int(rng.integers(2, 12))   # (few levels each: E stays small)
    for kind in kinds:
        if kind == "dense":
            k = int(rng.choice([1, 3, 5, 8, 10, 11, 12, 16, 17, 24, 33, 50, 64, 100, 128, 130]))
            X = rng.standard_normal((n, k)).astype(dtype)
            if rng.random() < 0.5:
                X = np.asfortranarray(X)
            blocks.append(tm.DenseMatrix(X))
            dense_parts.append(X.astype(np.float64))
        elif kind == "sparse":
            m = int(rng.choice([1, 5, 33, 128, 200, 513, 1100, 2100]))
            dens = float(rng.choice([0.0, 0.02, 0.05, 0.12, 0.4] if m < 1000 else [0.0008, 0.004, 0.02]))
            S = sps.random(n, m, density=dens, format="csc", random_state=rng).astype(dtype)
            blocks.append(tm.SparseMatrix(S))
            dense_parts.append(S.toarray().astype(np.float64))
        else:
            ncat = int(rng.choice([2, 3, 5, 12, 40] if kind == "cat_small" else [1, 2, 5, 40, 300, 700, 5000]))
            drop = bool(rng.random() < 0.4)
            codes = rng.integers(0, ncat, n)
            missing = rng.random() < 0.3
            if missing:
                codes = np.where(rng.random(n) < 0.1, -1, codes)
            blocks.append(tm.CategoricalMatrix(codes, categories=np.arange(ncat), drop_first=drop,
                                               dtype=dtype, cat_missing_method="zero" if missing else "fail"))
            oh = np.zeros((n, ncat))
            ok = codes >= 0
            oh[np.nonzero(ok)[0], codes[ok]] = 1.0
            dense_parts.append(oh[:, int(drop):])
    E = np.hstack(dense_parts) if dense_parts else np.zeros((n, 0))
    keep = [b for b, p in zip(blocks, dense_parts) if p.shape[1] > 0]
    if not keep:
        return None, None
    X = tm.SplitMatrix(keep) if len(keep) > 1 else keep[0]
    return X, E


@pytest.mark.parametrize("seed", range(int(os.environ.get("TM_FUZZ_CASES", "32"))))
def test_random_split_products(seed):
    import tabmat_amd as tm

    rng = np.random.default_rng(1000 + seed)
    dtype = np.float64 if seed % 4 else np.float32
    X, E = _random_split(rng, dtype)
    if X is None:
        pytest.skip("degenerate draw")
    n, p = E.shape
    tol = 1e-10 if dtype == np.float64 else 5e-4
    d = rng.random(n).astype(dtype)
    d[rng.random(n) < 0.1] = 0.0
    v = rng.standard_normal(p).astype(dtype)
    w = rng.standard_normal(n).astype(dtype)
    d64, v64, w64 = d.astype(np.float64), v.astype(np.float64), w.astype(np.float64)

    def close(a, b):
        a = np.asarray(a.toarray() if sps.issparse(a) else a, dtype=np.float64)
        scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
        assert a.shape == b.shape
        assert float(np.abs(a - b).max()) / scale < tol

    def close_sand(a, b):
        """close() + entry by entry at the natural scale sqrt(S_ii S_jj) (d >= 0 here): a wrong small block
        cannot hide under the largest one."""
        from _gpu_util import nat_err

        close(a, b)
        a = np.asarray(a.toarray() if sps.issparse(a) else a, dtype=np.float64)
        assert nat_err(a, b) < (1e-10 if dtype == np.float64 else 2e-3)

    close_sand(X.sandwich(d), E.T @ (d64[:, None] * E))
    close(X.matvec(v), E @ v64)
    close(X.transpose_matvec(w), E.T @ w64)
    rows = np.sort(rng.choice(n, size=max(1, n // 2), replace=False)).astype(np.int32)
    cols = np.sort(rng.choice(p, size=max(1, (2 * p) // 3), replace=False)).astype(np.int32)
    Er = E[np.ix_(rows, cols)]
    close_sand(X.sandwich(d, rows, cols), Er.T @ (d64[rows, None] * Er))
    close(X.matvec(v, cols), E[:, cols] @ v64[cols])
    close(X.transpose_matvec(w, rows, cols), Er.T @ w64[rows])
    # a narrow selection (below the share from which the unrestricted product + selection is used)
    few_c = np.sort(rng.choice(p, size=max(1, p // 5), replace=False)).astype(np.int32)
    Ec = E[:, few_c]
    close_sand(X.sandwich(d, None, few_c), Ec.T @ (d64[:, None] * Ec))
    close_sand(X.sandwich(d, rows, few_c), Ec[rows].T @ (d64[rows, None] * Ec[rows]))
    close(X.transpose_matvec(w, None, few_c), Ec.T @ w64)
    close(X.sandwich(d, cols=np.arange(p)), E.T @ (d64[:, None] * E))
    # a short row list (row-list kernels: cost proportional to len(rows)), with repeats allowed
    few = rng.choice(n, size=max(1, n // 9), replace=False).astype(np.int64)
    Ef = E[few]
    close_sand(X.sandwich(d, few), Ef.T @ (d64[few, None] * Ef))
    if isinstance(X, tm.SplitMatrix):
        Xd = X.to_device()
        close_sand(Xd.sandwich(d), E.T @ (d64[:, None] * E))
        # the standardized view on the same blocks (one pass for the inner sandwich and X' d)
        shift = rng.standard_normal(p)
        mult = rng.random(p) + 0.5
        Es = E * mult + shift
        got = tm.StandardizedMatrix(X, shift, mult).sandwich(d, rows, cols)
        want = Es[np.ix_(rows, cols)].T @ (d64[rows, None] * Es[np.ix_(rows, cols)])
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(np.asarray(got) - want).max()) / scale < (tol if dtype == np.float64 else 5e-3)
        # matvec / transpose_matvec of the standardized view, device vectors, with a selection
        import torch
        Sm = tm.StandardizedMatrix(X, shift, mult)
        vd = torch.as_tensor(v, device="cuda")
        wd = torch.as_tensor(w, device="cuda")
        mtol = tol if dtype == np.float64 else 5e-3
        got = Sm.matvec(vd, cols=cols).cpu().numpy()
        want = Es[:, cols] @ v64[cols]
        assert float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max())) < mtol
        got = Sm.transpose_matvec(wd, rows=rows, cols=few_c).cpu().numpy()
        want = Es[np.ix_(rows, few_c)].T @ w64[rows]
        assert float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max())) < mtol
        # ... and under the narrow selection (dense-block form of the selected columns)
        got = tm.StandardizedMatrix(X, shift, mult).sandwich(d, rows, few_c)
        Esf = Es[np.ix_(rows, few_c)]
        want = Esf.T @ (d64[rows, None] * Esf)
        scale = max(1.0, float(np.abs(want).max()))
        assert float(np.abs(np.asarray(got) - want).max()) / scale < (tol if dtype == np.float64 else 5e-3)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TM_FUZZ_STD_CASES", "24"))))
def test_random_standardized_sandwich(seed):
    """Round 5: StandardizedMatrix.sandwich over random designs with UNCENTRED dense columns (mean up to 300 standard
    deviations), random shift / mult (also ones that are NOT the columns' means: the centring is algebra, not
    statistics), rows / cols restrictions, numpy and device vectors -- against float64 dense algebra on the
    explicitly standardized matrix, entry by entry at the natural scale.  float64 designs only (float32 keeps the
    reference's formula)."""
    import torch

    import tabmat_amd as tm

    rng = np.random.default_rng(7000 + seed)
    X, E = None, None
    while X is None:
        X, E = _random_split(rng, np.float64)
    n, p = E.shape
    # move the dense columns off zero: x -> mean + x (in the matrix AND its dense image)
    mats = X.matrices if isinstance(X, tm.SplitMatrix) else [X]
    idxs = X.indices if isinstance(X, tm.SplitMatrix) else [np.arange(p)]
    new = []
    for mb, ix in zip(mats, idxs):
        if isinstance(mb, tm.DenseMatrix):
            mu = rng.choice([0.0, 3.0, -40.0, 300.0], size=mb.shape[1])
            A = mb.toarray() + mu[None, :]
            E[:, ix] = A
            new.append(tm.DenseMatrix(np.asfortranarray(A) if A.flags["F_CONTIGUOUS"] and A.ndim == 2 and rng.random() < 0.5 else A))
        else:
            new.append(mb)
    X = tm.SplitMatrix(new, [np.asarray(i) for i in idxs]) if isinstance(X, tm.SplitMatrix) else new[0]
    if seed % 3 == 0:                   # the statistics glum would use
        w = rng.random(n) + 0.1
        w /= w.sum()
        std = X.standardize(w, True, bool(seed % 2))[0]
    else:                               # arbitrary shift / mult
        mult = rng.uniform(0.2, 3.0, p) if seed % 2 else None
        shift = rng.standard_normal(p) * rng.choice([0.0, 1.0, 50.0], size=p)
        std = tm.StandardizedMatrix(X, shift, mult)
    Z = E * (std.mult[None, :] if std.mult is not None else 1.0) + std.shift[None, :]
    d = rng.random(n)
    d[rng.random(n) < 0.1] = 0.0
    rows = np.sort(rng.choice(n, max(1, n // 2), replace=False)) if seed % 4 == 1 and n > 1 else None
    cols = np.sort(rng.choice(p, max(1, int(p * rng.choice([0.2, 0.7]))), replace=False)) if seed % 4 >= 2 else None
    Zr = Z[rows if rows is not None else slice(None)][:, cols if cols is not None else slice(None)]
    dr = d[rows] if rows is not None else d
    want = (Zr.T * dr) @ Zr
    for dd in (d, torch.from_numpy(d).cuda()):
        got = std.sandwich(dd, rows=rows, cols=cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        # (the float64 reference itself carries eps * (mean / std) per entry of Z: 1e-9 at the natural scale)
        dg = np.sqrt(np.abs(np.diag(want)))
        den = np.outer(dg, dg)
        err = np.abs(got - want)
        ok = np.where(den > 0, err <= 1e-9 * den + 1e-12 * (np.abs(want).max() + 1.0), err <= 1e-12 * (np.abs(want).max() + 1.0))
        assert ok.all(), (float((err / np.where(den > 0, den, 1.0)).max()), rows is not None, cols is not None)
