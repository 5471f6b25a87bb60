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
                     j_drop_first=False, hot=None):
    """ext/split.pyx:83-111.  hot: an upper bound of the rows any one cell of the table collects
    (None = unknown); a large, evenly filled table then takes global atomics instead of dozens of
    LDS-tiled passes over the codes (tm_cat_cat_sandwich_atomic_*)."""
    if i_ncol == 0 or j_ncol == 0 or (rows is not None and D.nlen(rows) == 0):
        return D.zeros((i_ncol, j_ncol), d.dtype)
    res = D.out_buf((i_ncol, j_ncol), d.dtype)
    fn = "tm_cat_cat_sandwich_atomic_" if (hot is not None and hot <= 4096) else "tm_cat_cat_sandwich_"
    call(fn + D.fsuf(d), D.p(i_indices), D.p(j_indices), i_indices.numel(),
         D.p(d), D.p(rows), D.nlen(rows), i_ncol, j_ncol, int(i_drop_first), int(j_drop_first),
         D.p(res), D.stream_ptr())
    return res


CAT_CAT_SORTED_MIN_ROWS = 500_000      # (below: 19 ps per row saved do not pay for a static twin of the pair)


def cat_cat_sorted_pays(n, i_ncol, j_ncol, itemsize=8) -> bool:
    """The level-sorted table kernel (tm_cat_cat_sandwich_sorted_*) instead of one device-scope atomic per row
    (23 G/s) or one pass over the codes per LDS tile of the table: when the table needs several tiles, a row of it
    fits one, and the rows pay for the static sorted twin (12 bytes per row)."""
    if not (n >= CAT_CAT_SORTED_MIN_ROWS and 0 < j_ncol * 8 <= 128 * 1024 and i_ncol * j_ncol < 2**31 - 1):
        return False
    # measured per row: the LDS-tiled kernel ~3 ps per PASS (16 coalesced bytes; one pass per tile of levels), global
    # atomics 43 ps, this kernel 24 ps (12 coalesced bytes + a 64-byte sector for the gathered weight): it pays from
    # eight tiles on (BASELINE configs[3]'s 256 x 96 table is two tiles and stays with the tiled kernel)
    ti = (128 * 1024) // (8 * j_ncol)
    return -(-i_ncol // ti) >= 8


def sandwich_cat_cat_sorted(twin, i_ncol, j_ncol, d):
    """twin = (ci_sorted, cj_sorted, perm, lptr) of the pair (CategoricalMatrix._sorted_pair)."""
    ci_s, cj_s, perm, lptr = twin
    res = D.out_buf((i_ncol, j_ncol), d.dtype)
    call("tm_cat_cat_sandwich_sorted_" + D.fsuf(d), D.p(ci_s), D.p(cj_s), D.p(perm), D.p(lptr), int(perm.numel()),
         D.p(d), i_ncol, j_ncol, D.p(res), D.stream_ptr())
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
    cats: list of (block id, n_cols).  A bundle is a rectangle A x B of categoricals whose tile
    [sum of A's levels] x [sum of B's levels] fits the LDS (csrc/cat_pairs.hip); the triangle of
    all pairs is cut recursively (halving the side with more levels) until every piece fits.
    pairs: (block i, block j, offset in the tables buffer, L_i, L_j, stride) with i <= j; for
    i == j the L_i diagonal entries lie `stride` apart."""

    N_WG = 512          # workgroups dealt to the bundles (two rounds on 256 CUs)

    def __init__(self, cats, pos_arrays, diag_only=False):
        """diag_only: only the diagonals = the weighted histograms of all categoricals in one pass
        (SplitMatrix.transpose_matvec)."""
        from .._lib import lib
        import torch

        self.cat_ids = [c[0] for c in cats]
        k = len(cats)
        cap = int(lib().tm_multi_cat_pairs_max_bins())
        slots = int(lib().tm_multi_cat_pairs_max_slots())
        roww = int(lib().tm_multi_cat_pairs_row_words())
        L = [int(c[1]) for c in cats]
        bundles = []          # (A list, B list, mode)
        lone = []             # categoricals whose diagonal no triangle holds

        def tot(g):
            return sum(L[a] for a in g)

        def halves(g):
            h = len(g) // 2
            return g[:h], g[h:]

        def cross(A, B):
            if tot(A) * tot(B) <= cap and len(A) <= slots and len(B) <= slots:
                bundles.append((A, B, 0))
            elif len(A) == 1 and len(B) == 1:
                return                     # the pair's own table exceeds a tile: per-pair kernel
            elif len(B) == 1 or (len(A) > 1 and (tot(A) >= tot(B) or len(A) > slots)
                                 and not len(B) > slots):
                a1, a2 = halves(A)
                cross(a1, B)
                cross(a2, B)
            else:
                b1, b2 = halves(B)
                cross(A, b1)
                cross(A, b2)

        def tri(G):
            if tot(G) ** 2 <= cap and len(G) <= slots:
                bundles.append((G, G, 1))
            elif len(G) == 1:
                lone.append(G[0])
            else:
                g1, g2 = halves(G)
                tri(g1)
                tri(g2)
                cross(g1, g2)

        live = [a for a in range(k) if L[a] > 0]
        if diag_only:
            lone = list(live)
        elif live:
            tri(live)
        cur = []
        for a in lone:                     # diagonals only: one bin per level
            if L[a] > cap:
                continue
            if cur and (tot(cur) + L[a] > cap or len(cur) >= slots):
                bundles.append((cur, cur, 2))
                cur = []
            cur.append(a)
        if cur:
            bundles.append((cur, cur, 2))
        self.n_bundles = len(bundles)
        self.slots = max((max(len(A), len(B)) for A, B, _ in bundles), default=1)
        # workgroups in proportion to the work per row: one LDS atomic per table, one load per
        # categorical, the row's own bookkeeping
        def cost(A, B, mode):
            nt = len(A) * len(B) if mode == 0 else (len(A) * (len(A) + 1) // 2 if mode == 1 else len(A))
            nl = len(A) + (len(B) if mode == 0 else 0)
            return nt + 0.5 * nl + 1.0

        costs = [cost(*bd) for bd in bundles]
        n_wg = max(self.N_WG, self.n_bundles)
        share = [max(1, int(n_wg * c / max(sum(costs), 1e-9))) for c in costs]
        rows_w = np.zeros((max(self.n_bundles, 1), roww), dtype=np.int32)
        pstart = self.pstart = np.concatenate([[0], np.cumsum(L)]).astype(np.int64)
        desc, self.pairs, wg_map = [], [], []
        self.bins = max((tot(A) * (tot(B) if mode != 2 else 1) for A, B, mode in bundles), default=1)
        for y, (A, B, mode) in enumerate(bundles):
            width = tot(B) if mode != 2 else 1
            r = rows_w[y]
            r[0], r[1], r[2], r[3] = len(A), len(B), mode, width
            r[4], r[5], r[6] = len(wg_map), share[y], tot(A) * width
            wg_map += [y] * share[y]
            ra = np.concatenate([[0], np.cumsum([L[a] for a in A])]).astype(np.int64)
            cb = np.concatenate([[0], np.cumsum([L[b] for b in B])]).astype(np.int64)
            for s, a in enumerate(A):
                r[8 + 2 * s], r[9 + 2 * s] = a, ra[s] * width
            for s, b in enumerate(B):
                r[8 + 2 * slots + 2 * s], r[9 + 2 * slots + 2 * s] = b, (cb[s] if mode != 2 else 0)
            base = y * self.bins
            for s, a in enumerate(A):
                for t, b in enumerate(B):
                    if (mode == 1 and t < s) or (mode == 2 and t != s):
                        continue
                    if a == b:
                        off = base + ra[s] * width + (cb[s] if mode == 1 else 0)
                        stride = width + 1 if mode == 1 else 1
                        desc.append([off, L[a], L[a], stride, pstart[a], pstart[a], 1, 0])
                    else:
                        i, j = (a, b)
                        off = base + ra[s] * width + cb[t]
                        stride = width
                        desc.append([off, L[i], L[j], stride, pstart[i], pstart[j], 0, 0])
                    self.pairs.append((self.cat_ids[a], self.cat_ids[b], int(off), L[a], L[b], int(stride)))
        self.n_wg = len(wg_map)
        self.n_pairs = len(desc)
        # host copies (the layout is checked on the CPU: tests/test_abi_and_host_logic.py)
        self.h_bundles = rows_w
        self.h_wg_map = np.asarray(wg_map if wg_map else [0], dtype=np.int32)
        self.h_desc = np.asarray(desc, dtype=np.int64).reshape(-1, 8)
        if pos_arrays is None:             # layout only
            self.covered = {(min(i, j), max(i, j)) for i, j, *_ in self.pairs}
            return
        self.bundles = D.to_dev(rows_w.reshape(-1))
        self.wg_map = D.to_dev(self.h_wg_map)
        self.desc = D.to_dev(self.h_desc)
        self.pos = torch.cat([p.to(torch.int64) for p in pos_arrays]) if pos_arrays else \
            D.zeros((0,), torch.int64)
        self.covered = {(min(i, j), max(i, j)) for i, j, *_ in self.pairs}
        self._cat_tab = (None, None)

    def cat_tab(self, cats):
        """Device [k][4] int64 {codes pointer, first kept code, first entry in pos, 0}; rebuilt when
        a pointer changes."""
        key = tuple((c[0].data_ptr(), bool(c[2])) for c in cats)
        if self._cat_tab[0] != key:
            tab = np.asarray([[c[0].data_ptr(), int(bool(c[2])), int(self.pstart[a]), 0]
                              for a, c in enumerate(cats)], dtype=np.int64)
            self._cat_tab = (key, D.to_dev(tab.reshape(-1)))
        return self._cat_tab[1]


def multi_cat_pairs(plan: CatPairsPlan, cats, d, rows, out, vector=False, pos=None):
    """All bundled categorical x categorical tables + diagonals in one pass, scattered into the
    float64 (p, p) `out`; returns the tables buffer (float64, [n_bundles * bins]).
    vector: `out` is a float64 VECTOR over the columns and only diagonals are written
    (out[pos] = histogram: a diag_only plan).  pos: positions to use instead of the plan's (-1 =
    level not selected: nothing is written for it)."""
    import torch

    tables = D.out_buf((max(plan.n_bundles, 1) * plan.bins,), torch.float64)
    if plan.n_pairs == 0:
        return tables
    if rows is not None and rows.numel() == 0:
        # (an empty device tensor has a NULL pointer, which the C ABI reads as "all rows")
        return tables.zero_()
    nrows = int(cats[0][0].numel())
    call(f"tm_multi_cat_pairs_{D.fsuf(d)}", D.p(plan.cat_tab(cats)), nrows, D.p(d), D.p(rows),
         D.nlen(rows), D.p(plan.bundles), plan.n_bundles, D.p(plan.wg_map), plan.n_wg, plan.slots,
         plan.bins, D.p(plan.desc), plan.n_pairs, D.p(plan.pos if pos is None else pos), D.p(tables),
         D.p(out), 0 if (vector or out is None) else out.shape[0], D.stream_ptr())
    return tables


def multi_cat_matvec(plan: CatPairsPlan, cats, v, out):
    """out[k] += sum over the plan's categoricals of v[position of the row's level] (one pass over
    all codes; v: the FULL coefficient vector of the split matrix)."""
    D.same_float("multi_cat_matvec", v, out)
    call(f"tm_multi_cat_matvec_{D.fsuf(v)}", D.p(plan.cat_tab(cats)), len(cats), D.p(plan.pos), D.p(v),
         int(cats[0][0].numel()), D.p(out), D.stream_ptr())
    return out


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


def _col_maps(self):
    """(block of column c, local index of column c) for a SplitMatrix whose blocks list their
    columns in increasing order and cover every column once; else None.  Cached on the matrix."""
    maps = self.__dict__.get("_col_maps", False)
    if maps is False:
        maps = None
        inds = [np.asarray(i) for i in self.indices]
        p = int(sum(i.size for i in inds))
        if all(i.size < 2 or bool((i[1:] > i[:-1]).all()) for i in inds) and \
                all(i.size == 0 or (i.min() >= 0 and i.max() < p) for i in inds):
            block_of = np.full(p, -1, dtype=np.int64)
            local_of = np.zeros(p, dtype=np.int32)
            for j, i in enumerate(inds):
                block_of[i] = j
                local_of[i] = np.arange(i.size, dtype=np.int32)
            if p == 0 or block_of.min() >= 0:
                maps = (block_of, local_of)
        self.__dict__["_col_maps"] = maps
    return maps


def split_col_subsets(self, cols: np.ndarray):
    """ext/split.pyx:157-209: host-side index bookkeeping (p-sized), same outputs:
    (subset_cols_indices, subset_cols, n_cols)."""
    cols = np.asarray(cols, dtype=np.int32)
    n_blocks = len(self.indices)
    maps = _col_maps(self)
    if maps is not None and (cols.size < 2 or bool((cols[1:] > cols[:-1]).all())):
        # strictly increasing selection, every block's own indices increasing (the case the
        # reference's merge loop is written for): the same lists from two table lookups
        block_of, local_of = maps
        b, loc = block_of[cols], local_of[cols]
        order = np.argsort(b, kind="stable").astype(np.int32)
        ends = np.cumsum(np.bincount(b, minlength=n_blocks))
        starts = np.concatenate([[0], ends[:-1]])
        return ([order[s:e] for s, e in zip(starts, ends)],
                [loc[order[s:e]] for s, e in zip(starts, ends)], len(cols))
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
    multi_cat_dense_wide_kernel): C-ordered operand with 16-byte aligned rows, <= 8 categoricals,
    stacked tile of DOUBLES [sum(n_cols)][32] plus the per-wave scratch within 150 KB of LDS."""
    fb = mat_j.buf.element_size()
    vec = 2
    total = sum(int(c[1]) for c in cats)
    tile = (total * 16 * vec * 8 + 15) // 16 * 16
    lds = tile + 16 * 32 * (fb + 4 * len(cats))
    return (not mat_j.order_f and 1 <= len(cats) <= 8 and mat_j.m >= vec and mat_j.m % vec == 0
            and mat_j.buf.data_ptr() % 16 == 0 and lds <= 150 * 1024)


def multi_cat_dense_tile_ok(cats, mat_j: DenseDev) -> bool:
    """True when the generic LDS-tile kernel of tm_multi_cat_dense_sandwich_* (any width, order and
    alignment; tile [sum(n_cols)][TJ] doubles, TJ a power of two) covers a NARROW operand in at most two
    column parts -- e.g. the reference's benchmark design dense_cat (2 x 1000 levels, 5 dense columns:
    0.11 ms against 1.96 ms for the one-hot gather that F-ordered / unaligned operands took before)."""
    total = sum(int(c[1]) for c in cats)
    m = mat_j.m
    if total == 0 or m == 0 or not 1 <= len(cats) <= 16:
        return False
    tj = 64
    while tj > 1 and 8 * total * tj > 128 * 1024:        # HIST_LDS_MAX of csrc/cat.hip
        tj >>= 1
    while tj > 1 and tj // 2 >= m:
        tj >>= 1
    return 8 * total * tj <= 128 * 1024 and (m + tj - 1) // tj <= 2


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


PACKED_CODES = True       # categorical x sparse on the entry twin: the codes of <= 3 categoricals as one word per row


def pack_codes(cats):
    """uint32 device vector for tm_multi_cat_sparse_sandwich_entp_* (None when the set does not qualify: more than
    3 categoricals or 1022 stacked levels).  Depends on the codes only: the caller caches it."""
    import torch

    total = sum(int(c[1]) for c in cats)
    if not PACKED_CODES or len(cats) > 3 or total >= 1023 or len(cats) == 0:
        return None
    codes, ncols, drop, n = _cat_args(cats)
    nrow = int(cats[0][0].numel())
    out = torch.empty((nrow,), dtype=torch.int32, device=cats[0][0].device)
    call("tm_multi_cat_pack_codes", codes, ncols, drop, n, nrow, D.p(out), D.stream_ptr())
    return out


def multi_cat_sparse_sandwich_ent(cats, d, E, packed=None):
    """The same on the ENTRY twin of the sparse block (E: SlabEnt, the stream of the sparse x dense kernel of
    round 4): stacked [sum(n_cols) x E.m]; no slab-form twin needed.  packed: the result of pack_codes(cats)."""
    total = sum(int(c[1]) for c in cats)
    if total == 0 or E.m == 0 or E.n == 0:
        return D.zeros((total, E.m), E.vals.dtype)
    res = D.out_buf((total, E.mk), E.vals.dtype)
    codes, ncols, drop, n = _cat_args(cats)
    D.same_float("multi_cat_sparse_sandwich_ent", E.vals, d)
    if packed is not None:
        call(f"tm_multi_cat_sparse_sandwich_entp_{D.fsuf(E.vals)}", codes, ncols, drop, n, E.n, D.p(d),
             D.p(E.vals), D.p(E.meta), D.p(E.bstart), E.n_slots(), E.mk, D.p(packed), D.p(res), D.stream_ptr())
        return res[:, E.inv]
    call(f"tm_multi_cat_sparse_sandwich_ent_{D.fsuf(E.vals)}", codes, ncols, drop, n, E.n, D.p(d),
         D.p(E.vals), D.p(E.meta), D.p(E.bstart), E.n_slots(), E.mk, D.p(res), D.stream_ptr())
    return res[:, E.inv]      # kernel columns -> the block's columns


def multi_cat_sparse_sandwich_rows(cats, d, A, rows):
    """The fused categorical x sparse cross terms over a short row list: cost proportional to
    len(rows) (the reference works on self[rows], categorical_matrix.py:825-838).  A: CsrDev (its
    chunk-major twin and the row table of the row-list K2 are used); rows: int32 device tensor of
    row ids (a repeated id counts per occurrence, as X[rows] does in the reference); d: the full-length weight
    vector."""
    from . import sparse as xs

    total = sum(int(c[1]) for c in cats)
    if total == 0 or A.m == 0 or D.nlen(rows) == 0 or A.data.numel() == 0:
        return D.zeros((total, A.m), A.data.dtype)
    res = D.out_buf((total, A.m), A.data.dtype)
    codes, ncols, drop, n = _cat_args(cats)
    D.same_float("multi_cat_sparse_sandwich_rows", A.data, d)
    cm_data, cm_c8, ranges, r32, d_sel = xs._row_table(A, rows, d, False)
    call(f"tm_multi_cat_sparse_sandwich_rows_u8_{D.fsuf(A.data)}", codes, ncols, drop, n, D.p(cm_data),
         D.p(cm_c8), D.p(ranges), D.p(r32), int(r32.numel()), A.m, D.p(d_sel), D.p(res), D.stream_ptr())
    return res
