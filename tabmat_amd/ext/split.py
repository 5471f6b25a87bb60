"""Mirror of tabmat.ext.split (reference: src/tabmat/ext/split.pyx)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _device as D
from .._lib import call
from ._types import CsrDev, DenseDev, SlabCsc


def sandwich_cat_dense(i_indices, i_ncol, d, mat_j: DenseDev, rows, j_cols, drop_first=False):
    """ext/split.pyx:32-80."""
    n_j = mat_j.m if j_cols is None else D.nlen(j_cols)
    res = D.zeros((i_ncol, n_j), mat_j.dtype)
    n_rows = i_indices.numel() if rows is None else D.nlen(rows)
    if d.numel() == 0 or n_rows == 0 or n_j == 0 or i_ncol == 0:  # ext/split.pyx:51-52
        return res
    D.same_float("sandwich_cat_dense", mat_j.buf, d)
    call(f"tm_cat_dense_sandwich_{D.fsuf(mat_j.buf)}", D.p(i_indices), i_indices.numel(), i_ncol,
         int(drop_first), D.p(d), D.p(rows), D.nlen(rows), D.p(mat_j.buf), mat_j.m, mat_j.order_f,
         D.p(j_cols), D.nlen(j_cols), D.p(res), D.stream_ptr())
    return res


def sandwich_cat_cat(i_indices, j_indices, i_ncol, j_ncol, d, rows, i_drop_first=False,
                     j_drop_first=False):
    """ext/split.pyx:83-111."""
    if i_ncol == 0 or j_ncol == 0 or (rows is not None and D.nlen(rows) == 0):
        return D.zeros((i_ncol, j_ncol), d.dtype)
    res = D.out_buf((i_ncol, j_ncol), d.dtype)
    call(f"tm_cat_cat_sandwich_{D.fsuf(d)}", D.p(i_indices), D.p(j_indices), i_indices.numel(),
         D.p(d), D.p(rows), D.nlen(rows), i_ncol, j_ncol, int(i_drop_first), int(j_drop_first),
         D.p(res), D.stream_ptr())
    return res


def sandwich_cat_sparse(i_indices, i_ncol, d, S: CsrDev, rows, cols, drop_first=False):
    """The kernel behind CategoricalMatrix._cross_sparse (categorical_matrix.py:825-838), which in
    the reference is a scipy.sparse product."""
    n_cols = S.m if cols is None else D.nlen(cols)
    res = D.zeros((i_ncol, n_cols), S.dtype)
    if i_ncol == 0 or n_cols == 0 or (rows is not None and D.nlen(rows) == 0):
        return res
    D.same_float("sandwich_cat_sparse", S.data, d)
    call(f"tm_cat_sparse_sandwich_{D.fsuf(S.data)}", D.p(i_indices), i_indices.numel(), i_ncol,
         int(drop_first), D.p(S.data), D.p(S.indices), D.p(S.indptr), S.m, D.p(d), D.p(rows),
         D.nlen(rows), D.p(cols), D.nlen(cols), D.p(res), D.stream_ptr())
    return res


def cat_dense_sandwich_sorted(plan, n_cols, d, mat_j: DenseDev):
    """ext/split.pyx:32-80 for a categorical with many levels: the rows grouped by level once
    (plan = CategoricalMatrix._det_plan()), one pass over the C-ordered dense block whatever the
    number of levels (csrc/cat_sorted.hip).  Returns (n_cols, mat_j.m)."""
    perm, bstart, n_blocks, cat_bptr = plan
    if n_cols == 0 or mat_j.m == 0:
        return D.zeros((n_cols, mat_j.m), mat_j.dtype)
    res = D.out_buf((n_cols, mat_j.m), mat_j.dtype)
    D.same_float("cat_dense_sandwich_sorted", mat_j.buf, d)
    call(f"tm_cat_dense_sandwich_sorted_{D.fsuf(mat_j.buf)}", D.p(perm), D.p(bstart), int(n_blocks),
         D.p(cat_bptr), int(n_cols), D.p(d), D.p(mat_j.buf), mat_j.m, D.p(res), D.stream_ptr())
    return res


def cat_dense_sorted_ok(mat_j: DenseDev) -> bool:
    vec = 16 // mat_j.buf.element_size()
    return (not mat_j.order_f and mat_j.m >= vec and mat_j.m % vec == 0
            and mat_j.buf.data_ptr() % 16 == 0)


def cat_sparse_sandwich_sorted(plan, n_cols, d, S: CsrDev):
    """The scipy product behind CategoricalMatrix._cross_sparse (categorical_matrix.py:825-838) for
    a categorical with many levels: rows grouped by level, an LDS row of doubles per block
    (csrc/cat_sorted.hip).  Returns (n_cols, S.m)."""
    perm, bstart, n_blocks, cat_bptr = plan
    if n_cols == 0 or S.m == 0:
        return D.zeros((n_cols, S.m), S.dtype)
    res = D.out_buf((n_cols, S.m), S.dtype)
    D.same_float("cat_sparse_sandwich_sorted", S.data, d)
    call(f"tm_cat_sparse_sandwich_sorted_{D.fsuf(S.data)}", D.p(perm), D.p(bstart), int(n_blocks),
         D.p(cat_bptr), int(n_cols), D.p(d), D.p(S.data), D.p(S.indices), D.p(S.indptr), S.m, D.p(res),
         D.stream_ptr())
    return res


class CatPairsPlan:
    """Packing of the categorical x categorical tables (and diagonals) of a SplitMatrix into
    LDS-sized bundles for tm_multi_cat_pairs_* -- static per matrix, built once.
    cats: list of (block id, n_cols); pairs listed as (block i, block j) with i <= j."""

    def __init__(self, cats, pos_arrays):
        from .._lib import lib
        import torch

        self.cat_ids = [c[0] for c in cats]
        k = len(cats)
        cap = int(lib().tm_multi_cat_pairs_max_bins())
        max_tab = int(lib().tm_multi_cat_pairs_max_tables())
        sizes = [int(c[1]) for c in cats]
        # greedy bundles over the tables that fit one tile (diagonals first: they are tiny)
        # (a table is bundled only when at least four of its size share a tile: a bundle with one
        # or two big tables is no better than the per-pair kernel, which has the whole chip)
        items = [(a, a, sizes[a]) for a in range(k) if sizes[a] <= cap] + \
                [(a, b, sizes[a] * sizes[b]) for b in range(k) for a in range(b)
                 if sizes[a] * sizes[b] <= cap // 4]
        bundles, fill = [[]], 0
        for a, b, sz in items:
            if sz == 0:
                continue
            if fill + sz > cap or len(bundles[-1]) >= max_tab:
                bundles.append([])
                fill = 0
            bundles[-1].append((a, b, fill, sz))
            fill += sz
        bundles = [bd for bd in bundles if bd]
        self.n_bundles = len(bundles)
        self.bins = max((bd[-1][2] + bd[-1][3] for bd in bundles), default=1)
        stride = 4 + 4 * max((len(bd) for bd in bundles), default=0)
        pl = np.zeros(4 + max(self.n_bundles, 1) * stride, dtype=np.int64)
        pl[0] = stride
        pstart = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        desc, self.pairs = [], []
        for y, bd in enumerate(bundles):
            row = 4 + y * stride
            mask = 0
            for t, (a, b, o, sz) in enumerate(bd):
                pl[row + 4 + 4 * t: row + 8 + 4 * t] = (a, b, o, sizes[b])
                mask |= (1 << a) | (1 << b)
                desc.append([y * self.bins + o, sizes[a], sizes[b], pstart[a], pstart[b], int(a == b)])
                self.pairs.append((self.cat_ids[a], self.cat_ids[b], y * self.bins + o, sizes[a], sizes[b]))
            pl[row] = mask
            pl[row + 1] = len(bd)
        self.pair_list = D.to_dev(pl.astype(np.uint32).view(np.int32))
        self.desc = D.to_dev(np.asarray(desc, dtype=np.int64).reshape(-1, 6))
        self.pos = torch.cat([p.to(torch.int64) for p in pos_arrays]) if pos_arrays else \
            D.zeros((0,), torch.int64)
        self.n_pairs = len(desc)
        self.covered = {(min(i, j), max(i, j)) for i, j, *_ in self.pairs}


def multi_cat_pairs(plan: CatPairsPlan, cats, d, rows, out):
    """All bundled categorical x categorical tables + diagonals in one pass, scattered into the
    float64 (p, p) `out`; returns the tables buffer (float64, [n_bundles * bins])."""
    import torch

    tables = D.out_buf((max(plan.n_bundles, 1) * plan.bins,), torch.float64)
    if plan.n_pairs == 0:
        return tables
    codes, ncols, drop, n = _cat_args(cats)
    nrows = int(cats[0][0].numel())
    call(f"tm_multi_cat_pairs_{D.fsuf(d)}", codes, ncols, drop, n, nrows, D.p(d), D.p(rows),
         D.nlen(rows), D.p(plan.pair_list), plan.n_bundles, plan.bins, D.p(plan.desc), plan.n_pairs,
         D.p(plan.pos), D.p(tables), D.p(out), out.shape[0], D.stream_ptr())
    return tables


def scatter_block(src, ri, ci, out, mirror=False, diag=False):
    """out[ri[a], ci[b]] = src[a, b] (+ transpose); diag: out[ri[a], ri[a]] += src[a].
    Device form of split_matrix.py:341-354."""
    p = out.shape[0]
    if diag:
        nr, nc = src.numel(), 1
    else:
        nr, nc = src.shape
    if nr == 0 or nc == 0:
        return
    call(f"tm_scatter_block_{D.fsuf(src)}", D.p(src), nr, nc, D.p(ri), D.p(ci), D.p(out), p,
         int(mirror), int(diag), D.stream_ptr())


def split_col_subsets(self, cols: np.ndarray):
    """ext/split.pyx:157-209: host-side index bookkeeping (p-sized), same outputs:
    (subset_cols_indices, subset_cols, n_cols)."""
    cols = np.asarray(cols, dtype=np.int32)
    n_blocks = len(self.indices)
    next_idx = [0] * n_blocks
    sub_idx = [[] for _ in range(n_blocks)]
    sub_cols = [[] for _ in range(n_blocks)]
    for i, c in enumerate(cols.tolist()):
        for j in range(n_blocks):
            ind = self.indices[j]
            k = next_idx[j]
            while k < len(ind) and ind[k] < c:
                k += 1
            next_idx[j] = k
            if k < len(ind) and ind[k] == c:
                sub_idx[j].append(i)
                sub_cols[j].append(k)
                next_idx[j] = k + 1
                break
    return (
        [np.array(s, dtype=np.int32) for s in sub_idx],
        [np.array(s, dtype=np.int32) for s in sub_cols],
        len(cols),
    )


def is_sorted(a) -> bool:
    """ext/split.pyx:211-217."""
    a = np.asarray(a)
    return bool(np.all(a[1:] >= a[:-1]))


MAX_FUSED_CATS = 16


def _cat_args(cats):
    """cats: list of (codes tensor, n_cols, drop_first) -> host arrays for the C ABI."""
    n = len(cats)
    codes = (C.c_void_p * n)(*[c[0].data_ptr() for c in cats])
    ncols = (C.c_int64 * n)(*[int(c[1]) for c in cats])
    drop = (C.c_int32 * n)(*[int(bool(c[2])) for c in cats])
    return codes, ncols, drop, n


def multi_cat_dense_wide_ok(cats, mat_j: DenseDev) -> bool:
    """True when tm_multi_cat_dense_sandwich_* takes its wide-load LDS-atomic kernel (cat.hip,
    multi_cat_dense_wide_kernel): C-ordered operand with 16-byte aligned rows, <= 4 categoricals,
    stacked tile of DOUBLES [sum(n_cols)][32] plus the per-wave scratch within 150 KB of LDS."""
    fb = mat_j.buf.element_size()
    vec = 2
    total = sum(int(c[1]) for c in cats)
    tile = (total * 16 * vec * 8 + 15) // 16 * 16
    lds = tile + 16 * 32 * (fb + 4 * len(cats))
    return (not mat_j.order_f and 1 <= len(cats) <= 4 and mat_j.m >= vec and mat_j.m % vec == 0
            and mat_j.buf.data_ptr() % 16 == 0 and lds <= 150 * 1024)


def multi_cat_dense_sandwich(cats, d, mat_j: DenseDev, rows=None):
    """All categorical x dense cross blocks in one pass over the dense block:
    returns the stacked [sum(n_cols) x mat_j.m] result (ext/split.pyx:32-80, fused over cats).
    rows (int32 device tensor): only those rows are read (wide-load path only, see
    multi_cat_dense_wide_ok)."""
    total = sum(int(c[1]) for c in cats)
    if total == 0 or mat_j.m == 0 or mat_j.n == 0 or (rows is not None and D.nlen(rows) == 0):
        return D.zeros((total, mat_j.m), mat_j.dtype)
    res = D.out_buf((total, mat_j.m), mat_j.dtype)
    codes, ncols, drop, n = _cat_args(cats)
    if rows is not None:
        D.same_float("multi_cat_dense_sandwich", mat_j.buf, d)
        call(f"tm_multi_cat_dense_sandwich_rows_{D.fsuf(mat_j.buf)}", codes, ncols, drop, n, mat_j.n,
             D.p(d), D.p(mat_j.buf), mat_j.m, D.p(rows), D.nlen(rows), D.p(res), D.stream_ptr())
        return res
    D.same_float("multi_cat_dense_sandwich", mat_j.buf, d)
    call(f"tm_multi_cat_dense_sandwich_{D.fsuf(mat_j.buf)}", codes, ncols, drop, n, mat_j.n,
         D.p(d), D.p(mat_j.buf), mat_j.m, mat_j.order_f, D.p(res), D.stream_ptr())
    return res


def multi_cat_sparse_sandwich(cats, d, S: SlabCsc):
    """All categorical x sparse cross blocks in one pass over the slab-form sparse block:
    stacked [sum(n_cols) x S.m] (the scipy product of categorical_matrix.py:825-838, fused)."""
    total = sum(int(c[1]) for c in cats)
    if total == 0 or S.m == 0 or S.n == 0:
        return D.zeros((total, S.m), S.vals.dtype)
    res = D.out_buf((total, S.m), S.vals.dtype)
    codes, ncols, drop, n = _cat_args(cats)
    D.same_float("multi_cat_sparse_sandwich", S.vals, d)
    call(f"tm_multi_cat_sparse_sandwich_slab_{D.fsuf(S.vals)}", codes, ncols, drop, n, S.n, D.p(d),
         D.p(S.vals), D.p(S.koff), D.p(S.ecol), D.p(S.gptr), S.m, D.p(res), D.stream_ptr())
    return res
