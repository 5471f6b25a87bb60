"""Mirror of tabmat.ext.sparse (reference: src/tabmat/ext/sparse.pyx).  Only the CSR twin of
a sparse block is kept on the device; see include/tabmat_hip.h."""
from __future__ import annotations

import os


from .. import _device as D
from .._lib import call
from ._types import CsrDev, DenseDev, SlabCsc, SlabEll, SlabEnt, SlabLg


def _csr_args(A: CsrDev):
    return (D.p(A.data), D.p(A.indices), D.p(A.indptr), A.n, A.m)


def sparse_sandwich(A: CsrDev, d, rows, cols):
    """ext/sparse.pyx:17-77."""
    m = A.m if cols is None else D.nlen(cols)
    out = D.zeros((m, m), A.dtype)
    if m == 0 or (rows is not None and D.nlen(rows) == 0):
        return out
    D.same_float("sparse_sandwich", A.data, d)
    call(f"tm_sparse_sandwich_{D.fsuf(A.data)}", *_csr_args(A), D.p(d), D.p(rows), D.nlen(rows),
         D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def csr_dense_sandwich(A: CsrDev, B: DenseDev, d, rows, A_cols, B_cols):
    """ext/sparse.pyx:211-260."""
    nA = A.m if A_cols is None else D.nlen(A_cols)
    nB = B.m if B_cols is None else D.nlen(B_cols)
    nr = A.n if rows is None else D.nlen(rows)
    out = D.zeros((nA, nB), A.dtype)
    if nr == 0 or nA == 0 or nB == 0 or A.data.numel() == 0:  # ext/sparse.pyx:236-237
        return out
    D.same_float("csr_dense_sandwich", A.data, B.buf, d)
    call(f"tm_csr_dense_sandwich_{D.fsuf(A.data)}", *_csr_args(A), D.p(B.buf), B.m, B.order_f,
         D.p(d), D.p(rows), D.nlen(rows), D.p(A_cols), D.nlen(A_cols), D.p(B_cols),
         D.nlen(B_cols), D.p(out), D.stream_ptr())
    return out


CSR_U16 = os.environ.get("TABMAT_AMD_CSR_U16", "1") == "1"   # 16-bit column twin for the unrestricted matvec kernels
CSR_U16_MIN_NNZ = int(os.environ.get("TABMAT_AMD_CSR_U16_MIN_NNZ", "1000000"))   # (0: every block, for tests)


def _u16_ok(X: CsrDev, lds_bytes) -> bool:
    """The 16-bit column twin pays (and the kernel's LDS budget holds): a block of at most 65536 columns with enough
    entries that 2 of 12 bytes per entry matter; not a row-sliced view of odd parity (the pair loads need the values
    and the columns to start at entries of the same parity)."""
    isz = X.data.element_size()
    return (CSR_U16 and 0 < X.m <= 65536 and int(X.data.numel()) >= CSR_U16_MIN_NNZ and lds_bytes
            and (X.data.data_ptr() // isz) % 2 == 0)


def csr_matvec(X: CsrDev, v, rows, cols, out=None):
    """ext/sparse.pyx:79-140 (csr_matvec_unrestricted with rows = cols = None)."""
    n_rows = X.n if rows is None else D.nlen(rows)
    if out is None:
        out = D.zeros((n_rows,), X.dtype)
    if n_rows == 0 or (cols is not None and D.nlen(cols) == 0):
        return out
    D.same_float("csr_matvec", X.data, v, out)
    if rows is None and cols is None and _u16_ok(X, X.data.element_size() * (X.m + 4096) <= 64 * 1024):
        call(f"tm_csr_matvec_u16_{D.fsuf(X.data)}", D.p(X.data), D.p(X.indices16()), D.p(X.indptr), X.n, X.m, D.p(v),
             D.p(out), D.stream_ptr())
        return out
    call(f"tm_csr_matvec_{D.fsuf(X.data)}", *_csr_args(X), D.p(v), D.p(rows), D.nlen(rows),
         D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def csc_rmatvec(X: CsrDev, v, rows, cols, out=None):
    """ext/sparse.pyx:142-199 (csc_rmatvec_unrestricted with rows = cols = None); evaluated on
    the CSR twin."""
    n_cols = X.m if cols is None else D.nlen(cols)
    if out is None:
        out = D.zeros((n_cols,), X.dtype)
    if n_cols == 0 or (rows is not None and D.nlen(rows) == 0):
        return out
    D.same_float("csc_rmatvec", X.data, v, out)
    if rows is None and cols is None and _u16_ok(X, 8 * (X.m + 1) + X.data.element_size() * 4096 <= 128 * 1024):
        call(f"tm_csr_rmatvec_u16_{D.fsuf(X.data)}", D.p(X.data), D.p(X.indices16()), D.p(X.indptr), X.n, X.m, D.p(v),
             D.p(out), D.stream_ptr())
        return out
    call(f"tm_csr_rmatvec_{D.fsuf(X.data)}", *_csr_args(X), D.p(v), D.p(rows), D.nlen(rows),
         D.p(cols), D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def csr_matvec_multi(X: CsrDev, V, rows, cols, transpose):
    """2-D operand (scipy.sparse products in the reference, sparse_matrix.py:252-254, 266-268):
    X[rows, cols] @ V[cols] -> (n_rows, K), or X[rows, cols].T @ V[rows] -> (n_cols, K)."""
    K = int(V.shape[1])
    n_cols = X.m if cols is None else D.nlen(cols)
    n_rows = X.n if rows is None else D.nlen(rows)
    out = D.zeros((n_cols if transpose else n_rows, K), X.dtype)
    if K == 0 or n_rows == 0 or n_cols == 0:
        return out
    D.same_float("csr_matvec_multi", X.data, V)
    Vc = V.contiguous()
    fn = "tm_csr_rmatvec_multi_" if transpose else "tm_csr_matvec_multi_"
    call(fn + D.fsuf(X.data), *_csr_args(X), D.p(Vc), K, D.p(rows), D.nlen(rows), D.p(cols),
         D.nlen(cols), D.p(out), D.stream_ptr())
    return out


def csr_dense_sandwich_slab(A: SlabCsc, B: DenseDev, d):
    """Fast path of ext/sparse.pyx:211-260 for an unrestricted product (B C- or F-ordered):
    slab-blocked gather kernel with static register accumulators (csrc/sparse.hip, K3)."""
    assert B.n == A.n
    out = D.zeros((A.m, B.m), A.vals.dtype)
    if A.m == 0 or B.m == 0 or A.n == 0:
        return out
    D.same_float("csr_dense_sandwich_slab", A.vals, B.buf, d)
    call(f"tm_csr_dense_sandwich_slab_{D.fsuf(A.vals)}", D.p(A.vals), D.p(A.koff), D.p(A.cnt),
         D.p(A.gptr), A.n, A.m, D.p(B.buf), B.m, B.order_f, D.p(d), D.p(out), D.stream_ptr())
    return out


def ell_supported(B: DenseDev) -> bool:
    """The static ELL gather kernel needs a C-ordered B with 16-byte aligned rows."""
    vec = 16 // B.buf.element_size()
    return (not B.order_f) and B.m % vec == 0 and B.m >= vec and B.buf.data_ptr() % 16 == 0


def csr_dense_sandwich_ell(A: SlabEll, B: DenseDev, d):
    """Fast path of ext/sparse.pyx:211-260 for an unrestricted product with a C-ordered B:
    interleaved-ELL twin, static iterations with skip masks (csrc/sparse.hip, K3 ELL)."""
    assert B.n == A.n and ell_supported(B)
    if A.m == 0 or B.m == 0 or A.n == 0:
        return D.zeros((A.m, B.m), A.vals.dtype)
    out = D.zeros((A.mk, B.m), A.vals.dtype)
    D.same_float("csr_dense_sandwich_ell", A.vals, B.buf, d)
    fn = "tm_csr_dense_sandwich_ellw_" if A.wide else "tm_csr_dense_sandwich_ell_"
    call(fn + D.fsuf(A.vals), D.p(A.vals), D.p(A.koff), D.p(A.gptr),
         A.n, A.mk, D.p(B.buf), B.m, D.p(d), D.p(out), D.stream_ptr())
    return out[A.inv]      # kernel rows are the density-sorted columns


def csr_dense_sandwich_lg(A: SlabLg, B: DenseDev, d, unc=None, want_colsum=False):
    """Fast path of ext/sparse.pyx:211-260 for an unrestricted product with a C-ordered B of
    more than 64 columns: lane-group twin, DPP-broadcast gather (csrc/sparse_lg.hip), on the
    compact stream (tm_csr_dense_sandwich_lgc_*) when the twin has been compacted.
    want_colsum: returns (out, A' d) -- the column sums ride along in the same pass."""
    assert B.n == A.n and ell_supported(B)
    dt = A.dtype
    if A.m == 0 or B.m == 0 or A.n == 0:
        z = D.zeros((A.m, B.m), dt)
        return (z, D.zeros((A.m,), dt)) if want_colsum else z
    out = D.out_buf((A.mk, B.m), dt)
    u = int(A.unc if unc is None else unc)
    cs = D.out_buf((A.mk,), dt) if want_colsum else None
    if A.cvals is not None:
        D.same_float("csr_dense_sandwich_lg", A.cvals, B.buf, d)
        call("tm_csr_dense_sandwich_lgc_" + D.fsuf(A.cvals), D.p(A.cvals), D.p(A.cmap), D.p(A.crec),
             D.p(A.xkoff), A.n, A.mk, D.p(B.buf), B.m, D.p(d), u, D.p(out), D.p(cs), D.stream_ptr())
    elif want_colsum:
        D.same_float("csr_dense_sandwich_lg", A.vals, B.buf, d)
        call("tm_csr_dense_sandwich_lg_xtd_" + D.fsuf(A.vals), D.p(A.vals), D.p(A.koff), D.p(A.xptr),
             D.p(A.xvals), D.p(A.xkoff), A.n, A.mk, D.p(B.buf), B.m, D.p(d), u, D.p(out), D.p(cs),
             D.stream_ptr())
    else:
        D.same_float("csr_dense_sandwich_lg", A.vals, B.buf, d)
        call("tm_csr_dense_sandwich_lg_" + D.fsuf(A.vals), D.p(A.vals), D.p(A.koff), D.p(A.xptr),
             D.p(A.xvals), D.p(A.xkoff), A.n, A.mk, D.p(B.buf), B.m, D.p(d), u, D.p(out), D.stream_ptr())
    # kernel rows are the density-sorted columns
    return (out[A.inv], cs[A.inv]) if want_colsum else out[A.inv]


def csr_dense_sandwich_ent(A: SlabEnt, B: DenseDev, d, want_colsum=False):
    """Fast path of ext/sparse.pyx:211-260 for an unrestricted product with a C-ordered B of more than 64
    columns: entry twin, accumulators picked at run time (csrc/sparse_ent.hip, round 4).
    want_colsum: returns (out, A' d) -- the column sums ride along in the same pass."""
    assert B.n == A.n and ell_supported(B)
    dt = A.dtype
    if A.m == 0 or B.m == 0 or A.n == 0:
        z = D.zeros((A.m, B.m), dt)
        return (z, D.zeros((A.m,), dt)) if want_colsum else z
    out = D.out_buf((A.mk, B.m), dt)
    cs = D.out_buf((A.mk,), dt) if want_colsum else None
    D.same_float("csr_dense_sandwich_ent", A.vals, B.buf, d)
    call("tm_csr_dense_sandwich_ent_" + D.fsuf(A.vals), D.p(A.vals), D.p(A.meta), D.p(A.bstart), A.n, A.mk,
         D.p(B.buf), B.m, D.p(d), D.p(out), D.p(cs), D.stream_ptr())
    return (out[A.inv], cs[A.inv]) if want_colsum else out[A.inv]


def _row_table(A: CsrDev, rows, d, as_set: bool):
    """(cm_data, cm_col8, ranges int32 [n_chunks, n_sel, 2], rows_sorted int32, d_sel): the
    {start, end} of every selected row in every column chunk, rows ascending (the sums do not
    depend on the order).  as_set: a repeated row counts once -- the reference's sparse_sandwich
    turns `rows` into a mask (ext/sparse.pyx:46-48) while its csr_dense_sandwich loops over the
    list and counts a repeated row twice (ext/sparse_helpers-tmpl.cpp:67-131)."""
    import weakref

    import torch

    cm_data, cm_c8, cptr = A.chunk_major()
    # the table depends on `rows` only: the self sandwich and the cross term of one call share it
    # (keyed on the tensor AND its version counter: a caller that refills a static index buffer in
    # place must not get the ranges of the old contents)
    cached = getattr(A, "_row_table", None)
    stamp = (rows.data_ptr(), rows._version, int(rows.numel()))
    if cached is not None and cached[0]() is rows and cached[4] == stamp:
        _, r64, dup, tabs, _ = cached
    else:
        r64 = torch.sort(rows.to(torch.int64)).values
        dup = bool((r64[1:] == r64[:-1]).any().item()) if r64.numel() > 1 else False
        tabs = {}
        A._row_table = (weakref.ref(rows), r64, dup, tabs, stamp)
    key = bool(as_set and dup)
    if key not in tabs:
        rr = torch.unique_consecutive(r64) if key else r64
        tabs[key] = (rr, torch.stack([cptr[:, rr], cptr[:, rr + 1]], dim=2).contiguous())
    rr, ranges = tabs[key]
    return cm_data, cm_c8, ranges, rr.to(torch.int32).contiguous(), d[rr].contiguous()


def sparse_sandwich_rows(A: CsrDev, d, rows):
    """A[rows]' diag(d[rows]) A[rows] at a cost proportional to len(rows) (ext/sparse.pyx:17-77
    with its `for k in rows` loop): the chunk-major K2 pipeline on the row table."""
    out = D.zeros((A.m, A.m), A.dtype)
    if A.m == 0 or D.nlen(rows) == 0 or A.data.numel() == 0:
        return out
    D.same_float("sparse_sandwich_rows", A.data, d)
    cm_data, cm_c8, ranges, r32, d_sel = _row_table(A, rows, d, True)
    cols = cm_c8 if K2B_U8 else A.chunk_cols32()        # (int32 columns: rebuilt per call, the A/B switch only)
    call(f"tm_sparse_sandwich_chunked_rows_{'u8_' if K2B_U8 else ''}{D.fsuf(A.data)}", D.p(cm_data),
         D.p(cols), D.p(ranges),
         int(r32.numel()), A.m, int(cm_data.numel()), D.p(d_sel), D.p(out), D.stream_ptr())
    return out


def csr_dense_sandwich_rows(A: CsrDev, B: DenseDev, d, rows):
    """A[rows]' diag(d[rows]) B[rows] at a cost proportional to len(rows) (ext/sparse.pyx:211-260,
    ext/sparse_helpers-tmpl.cpp:67-131): row-list kernel with an LDS tile per column chunk."""
    out = D.zeros((A.m, B.m), A.dtype)
    if A.m == 0 or B.m == 0 or D.nlen(rows) == 0 or A.data.numel() == 0:
        return out
    D.same_float("csr_dense_sandwich_rows", A.data, B.buf, d)
    cm_data, cm_c8, ranges, r32, d_sel = _row_table(A, rows, d, False)
    call(f"tm_csr_dense_sandwich_rows_u8_{D.fsuf(A.data)}", D.p(cm_data), D.p(cm_c8), D.p(ranges),
         int(r32.numel()), D.p(r32), D.p(d_sel), A.n, A.m, D.p(B.buf), B.m, B.order_f, D.p(out),
         D.stream_ptr())
    return out


def sparse_sandwich_chunked(A: CsrDev, d):
    """Unrestricted fast path of ext/sparse.pyx:17-77 on the chunk-major twin (K2)."""
    if A.m == 0 or A.n == 0:
        return D.zeros((A.m, A.m), A.dtype)
    out = D.out_buf((A.m, A.m), A.dtype)
    D.same_float("sparse_sandwich_chunked", A.data, d)
    cm_data, cm_c8, cptr = A.chunk_major()
    cols = cm_c8 if K2B_U8 else A.chunk_cols32()
    call(f"tm_sparse_sandwich_chunked_{'u8_' if K2B_U8 else ''}{D.fsuf(A.data)}", D.p(cm_data),
         D.p(cols), D.p(cptr),
         A.n, A.m, int(cm_data.numel()), D.p(d), D.p(out), D.stream_ptr())
    return out


# the block-list form of the unrestricted sparse self sandwich (csrc/sparse_blocks.hip)
K2_BLOCKS = True


def blocks_sandwich_pays(A: CsrDev) -> bool:
    """The block list is built for the regime the 8-slot chunked kernel serves (more than ~4.5
    nonzeros per row and 128-column chunk: rows regularly overflow 8 slots), when its 16 bytes per
    block fit: ~1.4 blocks per (row, tile)."""
    import torch

    if not K2_BLOCKS or A.n == 0 or A.m == 0 or A.n >= 2**29:
        return False
    nnz = int(A.data.numel())
    nch = -(-A.m // 128)
    if not (0 < nnz < 2**31) or nnz / (A.n * nch) <= 4.5 or nch > 32:
        return False
    est = 16 * 2 * A.n * nch * (nch + 1) // 2          # bytes, generous
    return est * 3 < torch.cuda.mem_get_info(A.data.device)[0] or bool(getattr(A, "_pb", None))


K2B_U8 = os.environ.get("TABMAT_AMD_K2B_U8", "1") == "1"     # byte columns for the block-list kernel's gathers


def sparse_sandwich_blocks(A: CsrDev, d):
    """ext/sparse.pyx:17-77, unrestricted, on the static block list (tm_sparse_sandwich_blocks_*)."""
    if A.m == 0 or A.n == 0:
        return D.zeros((A.m, A.m), A.dtype)
    out = D.out_buf((A.m, A.m), A.dtype)
    D.same_float("sparse_sandwich_blocks", A.data, d)
    cm_data, cm_c8, cptr = A.chunk_major()
    blocks, wg_tab, max_nb = A.pair_blocks(d12=None if K2B_U8 else False)     # (int32 columns: 16-byte list only)
    if int(blocks.shape[1]) == 3:            # 12-byte descriptors (always with byte columns)
        call(f"tm_sparse_sandwich_blocks_p12_{D.fsuf(A.data)}", D.p(cm_data), D.p(cm_c8), D.p(cptr), A.n, A.m,
             int(cm_data.numel()), D.p(blocks), D.p(wg_tab), int(wg_tab.shape[0]), int(max_nb), D.p(d),
             D.p(out), D.stream_ptr())
        return out
    if K2B_U8:
        call(f"tm_sparse_sandwich_blocks_u8_{D.fsuf(A.data)}", D.p(cm_data), D.p(cm_c8), D.p(cptr), A.n, A.m,
             int(cm_data.numel()), D.p(blocks), D.p(wg_tab), int(wg_tab.shape[0]), int(max_nb), D.p(d),
             D.p(out), D.stream_ptr())
        return out
    cm_ind = A.chunk_cols32()
    call(f"tm_sparse_sandwich_blocks_{D.fsuf(A.data)}", D.p(cm_data), D.p(cm_ind), D.p(cptr), A.n, A.m,
         int(cm_data.numel()), D.p(blocks), D.p(wg_tab), int(wg_tab.shape[0]), int(max_nb), D.p(d),
         D.p(out), D.stream_ptr())
    return out


def sparse_sandwich_pairs(A: CsrDev, d):
    """ext/sparse.pyx:17-77, unrestricted, for WIDE blocks: LDS tiles fed by the stream of a chunk's entries, one
    LDS atomic per pair of nonzeros of a row (tm_sparse_sandwich_pairs_*, csrc/sparse_pairs.hip)."""
    if A.m == 0 or A.n == 0:
        return D.zeros((A.m, A.m), A.dtype)
    out = D.out_buf((A.m, A.m), A.dtype)
    D.same_float("sparse_sandwich_pairs", A.data, d)
    _, _, cptr = A.chunk_major()
    if K2_PAIRS_PACKED and A.n < 2**25:
        rec = A.chunk_records_packed()
        call(f"tm_sparse_sandwich_pairs_pk_{D.fsuf(A.data)}", D.p(rec), D.p(cptr), A.n, A.m, int(A.data.numel()), D.p(d),
             D.p(out), D.stream_ptr())
        return out
    rec = A.chunk_records()
    call(f"tm_sparse_sandwich_pairs_{D.fsuf(A.data)}", D.p(rec), D.p(cptr), A.n, A.m, int(rec.shape[0]), D.p(d),
         D.p(out), D.stream_ptr())
    return out


def csc_dense_sandwich_sorted(A: CsrDev, B: DenseDev, d):
    """ext/sparse.pyx:211-260 for blocks with only a few nonzeros per row: column by column on
    the CSC form, a workgroup sums d[k] * A[k, j] * B[k, :] over (a block of) column j's entries
    (csrc/cat_sorted.hip).  Cost ~ one row of B per nonzero, independent of the width of A."""
    if A.m == 0 or B.m == 0:
        return D.zeros((A.m, B.m), A.dtype)
    rows, vals, bstart, n_blocks, col_bptr = A.csc_blocks()
    out = D.out_buf((A.m, B.m), A.dtype)
    D.same_float("csc_dense_sandwich_sorted", vals, B.buf, d)
    call(f"tm_csc_dense_sandwich_sorted_{D.fsuf(vals)}", D.p(rows), D.p(vals), D.p(bstart),
         int(n_blocks), D.p(col_bptr), A.m, D.p(d), D.p(B.buf), B.m, D.p(out), D.stream_ptr())
    return out


def sparse_sandwich_direct(A: CsrDev, d):
    """ext/sparse.pyx:17-77 for wide, very sparse blocks: one L2 atomic per pair
    (csrc/sparse_direct.hip); cost follows the pairs, not rows x tiles."""
    if A.m == 0 or A.n == 0:
        return D.zeros((A.m, A.m), A.dtype)
    out = D.out_buf((A.m, A.m), A.dtype)
    D.same_float("sparse_sandwich_direct", A.data, d)
    call(f"tm_sparse_sandwich_direct_{D.fsuf(A.data)}", D.p(A.data), D.p(A.indices), D.p(A.indptr),
         A.n, A.m, D.p(d), D.p(out), D.stream_ptr())
    return out


def direct_sandwich_pays(A: CsrDev) -> bool:
    """Cost model (measured, profiles/r2_microbench.txt): the tiled kernel pays ~15 ps per row and
    tile once rows have fewer than one nonzero per chunk, the direct one ~1 / 15e9 s per pair."""
    nnz, n, m = int(A.data.numel()), A.n, A.m
    if n == 0 or nnz == 0 or m <= 1024:
        return False
    nch = (m + 127) // 128
    per_row = nnz / n
    pairs = n * per_row * (per_row + 1.0) / 2.0 * 1.3          # (+ spread of the row lengths)
    if nch > 32:
        # beyond 32 column chunks only the generic tiled kernel is left, at ~0.3 ns per row and tile
        # (reference benchmark design 'sparse_wide', 40k x 10k at 1 %: 38 ms against 9.7 ms direct,
        # profiles/r3_bench_sparse_wide.json)
        return pairs / 15e9 < n * (nch * (nch + 1) / 2) * 0.3e-9
    if per_row / nch > 0.6:
        return False
    return pairs / 15e9 < n * (nch * (nch + 1) / 2) * 15e-12


# the pair-stream form of the unrestricted sparse self sandwich (csrc/sparse_pairs.hip): "auto" / "0" / "1"
K2_PAIRS = "auto"
K2_PAIRS_PACKED = os.environ.get("TABMAT_AMD_K2_PAIRS_PACKED", "1") == "1"    # packed 12- / 8-byte records (n < 2^25)
K2_PAIRS_MAX_M = 16384      # tm_sparse_sandwich_pairs_*: at most 128 column chunks
K2_PAIRS_MAX_NNZ = 2**31 - 4 * 16 * 64   # 32-bit entry positions + the kernel's look-ahead of 3 strides (KP_WAVES * 64)


def pairs_sandwich_pays(A: CsrDev) -> bool:
    """Cost model of the self-sandwich kernels for WIDE blocks (measured at 2M rows, profiles/r5_k2_pairs.txt):
      tiled (chunked kernel)   ~20 ps per (row, tile); block list ~1.15 ps per pair when rows hold > 4.5 entries
                               per 128-column chunk
      direct                   one L2 atomic per pair at ~21 G/s
      pairs (this kernel)      ~2.3 ps per visit of an entry (once per tile of its chunk's tile row), more when
                               the chunk's entries are thin over the rows (the gathers of d / the chunk pointers
                               span rows without entries: x (1.6 / k)^0.8 for k entries per row and chunk, at
                               most x 6) + 2.7 ps per pair + ~0.8 us per tile (LDS tile set-up and partials)."""
    if K2_PAIRS != "auto":
        return K2_PAIRS == "1" and A.m <= K2_PAIRS_MAX_M and 0 < int(A.data.numel()) < K2_PAIRS_MAX_NNZ
    nnz, n, m = int(A.data.numel()), A.n, A.m
    if n == 0 or nnz == 0 or m <= 512 or m > K2_PAIRS_MAX_M or nnz >= K2_PAIRS_MAX_NNZ:
        return False
    nch = (m + 127) // 128
    parts = nch * (nch + 1) / 2
    per_row = nnz / n
    k = per_row / nch
    pairs = n * per_row * (per_row + 2.0) / 2.0
    # (beyond 768 tiles every tile is one workgroup writing its tile once: ~0.1 us each, measured on the reference's
    # 'sparse_wide' design 40k x 10k @ 1 %: 1.63 ms, and on 40k x 16k @ 0.5 %: 2.71 ms)
    # (packed 12-byte records, n < 2^25: both per-entry terms x 0.75 -- 2.83 -> 2.12, 7.23 -> 5.54, 2.99 -> 2.18 ms)
    pk = 0.75 if (K2_PAIRS_PACKED and n < 2**25) else 1.0
    t_pairs = (nnz * (nch + 1) / 2.0 * 2.3e-12 * pk * min(6.0, max(1.0, (1.6 / k) ** 0.8)) + pairs * 2.7e-12 * pk
               + min(parts, 768) * 0.8e-6 + max(0.0, parts - 768) * 0.1e-6)
    t_direct = pairs / 21e9
    if nch > 32:
        t_tiled = n * parts * 0.3e-9      # only the generic tiled kernel is left (direct_sandwich_pays)
    else:
        # (chunked kernel at 1024 columns, 2M rows: 1.24 ms @ 1.25 %, 1.82 ms @ 2.5 % -- pair stream 0.56 / 1.59)
        t_tiled = pairs * 1.15e-12 if k > 4.5 else n * parts * 17e-12 + pairs * 0.8e-12
    # (a 10 % margin: 4096 columns @ 0.2 % over 2M rows is 2.9-3.0 ms here against 3.57 ms direct, modelled 3.34 / 3.98)
    return t_pairs < 0.9 * min(t_direct, t_tiled)


def transpose_square_dot_weights(A: CsrDev, weights):
    """ext/sparse.pyx:262-282: out[j] = sum_i w[i] * A[i, j]**2."""
    out = D.zeros((A.m,), A.dtype)
    if A.n == 0 or A.m == 0 or A.data.numel() == 0:
        return out
    D.same_float("transpose_square_dot_weights", A.data, weights)
    call(f"tm_csr_col_sq_{D.fsuf(A.data)}", D.p(A.data), D.p(A.indices), D.p(A.indptr), A.n, A.m,
         D.p(weights), D.p(out), D.stream_ptr())
    return out


def csr_densify_cols(A: CsrDev, colmap, T):
    """T[r, colmap[c]] += value for the stored entries of the selected columns (colmap: int32 device
    tensor [A.m], -1 = not selected); T: (n, ld) row-major device tensor, zeroed by the caller."""
    D.same_float("csr_densify_cols", A.data, T)
    call(f"tm_csr_densify_cols_{D.fsuf(T)}", D.p(A.data), D.p(A.indices), D.p(A.indptr), A.n,
         D.p(colmap), D.p(T), T.shape[1], D.stream_ptr())


def csc_densify_cols(rows, vals, seg, tcol, max_len, T):
    """The same from the CSC form (CsrDev.csc_blocks): seg int64 [n_sel, 2] entry ranges of the
    selected columns, tcol int32 [n_sel] their columns of T."""
    D.same_float("csc_densify_cols", vals, T)
    call(f"tm_csc_densify_cols_{D.fsuf(T)}", D.p(rows), D.p(vals), D.p(seg), D.p(tcol),
         int(tcol.numel()), int(max_len), D.p(T), T.shape[1], D.stream_ptr())
