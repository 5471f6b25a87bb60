"""SparseMatrix: a sorted CSC matrix on the host whose CSR twin is resident in HBM (reference:
/root/reference/src/tabmat/sparse_matrix.py).  Kernels: tabmat_amd/csrc/sparse.hip."""
from __future__ import annotations

import os

import numpy as np
import torch
from scipy import sparse as sps

from . import _device as D
from .ext import sparse as xs
from .ext._types import CsrDev, SlabCsc, SlabEll, SlabEnt, SlabLg
from .matrix_base import MatrixBase
from .util import (
    check_indexer,
    check_matvec_dimensions,
    check_matvec_out_shape,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    normalize_index,
    device_row_index,
    selects_all_columns,
)


# The interleaved-ELL twin pads every non-empty (slab, column group) to whole 64-slot iterations
# (x1.9 .. x2.5 at 5 % density); beyond this factor the compact slab stream is used instead.
ELL_MAX_PAD = 8.0
# K3's lane-group stream (the fallback of the entry twin since round 4) in its compact form: values of the real
# slots + a byte map, 2.7 instead of 7.7 GB at BASELINE configs[3] (False keeps the padded stream: tests only)
LG_COMPACT = True
# A row restriction with at most this fraction of the rows runs the row-list kernels (cost
# proportional to len(rows)); above it the full-pass kernels with a masked d are cheaper
# (scripts/dev/time_rows.py: break-even near one half for the self sandwich, one quarter for the
# sparse x dense term whose row-list form pays two LDS atomics per nonzero).
PART_NNZ = int(os.environ.get("TABMAT_AMD_PART_NNZ", str(2**31 - 2**24)))   # see split_matrix._parts
SORTED_K3_NNZ_PER_ROW = 6.0      # below: sparse x dense on the column-sorted kernel (wide blocks)
# largest column selection of a sparse self sandwich that is written out as a dense block (one syrk panel)
NARROW_COLS = int(os.environ.get("TABMAT_AMD_NARROW_COLS", "128"))
ROW_LIST_FRACTION = 0.5
ROW_LIST_FRACTION_K3 = 0.25


class SparseMatrix(MatrixBase):
    """Instantiated like scipy.sparse.csc_matrix (sparse_matrix.py:35-79), or from a ready
    CsrDev (device CSR twin) via SparseMatrix.from_device."""

    def __init__(self, input_array, shape=None, dtype=None, copy=False, column_names=None,
                 term_names=None):
        if isinstance(input_array, np.ndarray):
            if input_array.ndim == 1:
                input_array = input_array.reshape(-1, 1)
            elif input_array.ndim > 2:
                raise ValueError("Input array must be 1- or 2-dimensional")
        self._array = sps.csc_matrix(input_array, shape, dtype, copy)
        self.idx_dtype = max(self._array.indices.dtype, self._array.indptr.dtype)
        if self._array.indices.dtype != self.idx_dtype:
            self._array.indices = self._array.indices.astype(self.idx_dtype)
        if self._array.indptr.dtype != self.idx_dtype:
            self._array.indptr = self._array.indptr.astype(self.idx_dtype)
        if not self._array.has_canonical_format:
            # sorted, duplicate-free rows/columns: the chunked K2 kernel enumerates the pairs of a
            # row by entry position (the matrix itself is unchanged by summing duplicates)
            self._array.sum_duplicates()
        self._array_csr = None
        self._devblk = None
        self._slabblk = None
        self._ellblk = None
        self._shape = self._array.shape
        self._dtype = self._array.dtype
        self._init_names(column_names, term_names)

    def _init_names(self, column_names, term_names):
        width = self._shape[1]
        if column_names is not None and len(column_names) != width:
            raise ValueError(f"Expected {width} column names, got {len(column_names)}")
        if term_names is not None and len(term_names) != width:
            raise ValueError(f"Expected {width} term names, got {len(term_names)}")
        self._colnames = list(column_names) if column_names is not None else [None] * width
        self._terms = list(term_names) if term_names is not None else self._colnames

    @classmethod
    def from_device(cls, csr: CsrDev):
        """Wrap a CSR twin that already lives in HBM (no host copy).  The rows must be in
        canonical form (column indices ascending, no duplicates), as scipy's `sum_duplicates()`
        leaves them: the sparse self-sandwich enumerates the pairs of a row by entry position."""
        self = cls.__new__(cls)
        self._array = None
        self._array_csr = None
        self._devblk = csr
        self._slabblk = None
        self._ellblk = None
        self._shape = (csr.n, csr.m)
        self._dtype = np.dtype(np.float64 if csr.data.dtype == torch.float64 else np.float32)
        self.idx_dtype = np.dtype(np.int32)
        self._init_names(None, None)
        return self

    # ---- storage ------------------------------------------------------------------------
    def _host(self):
        if self._array is None:
            c = self._devblk
            csr = sps.csr_matrix((D.to_host(c.data), D.to_host(c.indices), D.to_host(c.indptr)),
                                 shape=self._shape)
            self._array = csr.tocsc()
            self._array.sort_indices()
        return self._array

    @property
    def array_csc(self):
        return self._host()

    @property
    def array_csr(self):
        """Host CSR twin (sparse_matrix.py:133-143)."""
        if self._array_csr is None:
            self._array_csr = self._host().tocsr(copy=False)
        return self._array_csr

    def _dev(self) -> CsrDev:
        if self._devblk is None:
            self._devblk = CsrDev.from_scipy(self.array_csr)
        return self._devblk

    def _values_finite(self) -> bool:
        """True when no stored value is inf / nan (checked once).  The gather kernel implements a
        row restriction as d = 0 on the excluded rows, and inf * 0 would leak a NaN from an
        excluded row; such blocks take the generic row-list kernel instead."""
        if getattr(self, "_finite", None) is None:
            self._finite = bool(torch.isfinite(self._dev().data).all().item())
        return self._finite

    def _slab(self) -> SlabCsc:
        """Slab-blocked column-major twin used by the sparse x dense gather kernel (built on
        first use; the reference likewise materialises its CSR twin lazily)."""
        if getattr(self, "_slabblk", None) is None:
            self._slabblk = SlabCsc.from_csr(self._dev())
        return self._slabblk

    def _ell(self, wide: bool = False) -> SlabEll:
        """Interleaved-ELL twin used by the static sparse x dense gather kernel (C-ordered B);
        wide: the 128-dense-column geometry used when B has more than 64 columns."""
        name = "_ellwblk" if wide else "_ellblk"
        if getattr(self, name, None) is None:
            twin = SlabEll.from_csr(self._dev(), wide=wide, max_pad=ELL_MAX_PAD)
            setattr(self, name, twin if twin is not None else False)
        twin = getattr(self, name)
        return twin if twin is not False else None      # None: too sparse, use the compact slab form

    def _lg(self) -> SlabLg:
        """Lane-group twin for the DPP-broadcast sparse x dense kernel (C-ordered B with more than
        64 columns); None when the block is too sparse or too dense for it."""
        if getattr(self, "_lgblk", None) is None:
            twin = SlabLg.from_csr(self._dev(), max_pad=ELL_MAX_PAD)
            if twin is not None and LG_COMPACT:
                twin.compact_()
            self._lgblk = twin if twin is not None else False
        return self._lgblk if self._lgblk is not False else None

    def _ent(self) -> SlabEnt:
        """Entry twin for the run-time-indexed sparse x dense kernel (round 4, csrc/sparse_ent.hip; C-ordered B
        with more than 64 columns); None when the block is so sparse that whole batches of 16 slots per
        (slab, column group) would exceed ELL_MAX_PAD x the nonzeros, or has 2^28 rows or more."""
        if getattr(self, "_entblk", None) is None:
            twin = SlabEnt.from_csr(self._dev(), max_pad=ELL_MAX_PAD)
            self._entblk = twin if twin is not None else False
        return self._entblk if self._entblk is not False else None

    def to_device(self, dense_width=None):
        """Upload and build the twins now (otherwise the first product does it).  dense_width:
        columns of the dense block this one will be crossed with (SplitMatrix.to_device passes it)
        -- selects the interleaved-ELL geometry to pre-build; None builds none."""
        self._dev().chunk_major()
        if xs.blocks_sandwich_pays(self._dev()):
            # the self sandwich's static block list: built here, not inside the first product (its builder reads
            # per-tile counts back to the host)
            self._dev().pair_blocks()
        ent = None
        if dense_width is not None and dense_width > 0:
            # (the entry twin only when the sparse x dense term will run on it: not for a few nonzeros per row in a
            # wide block, which takes the column-sorted kernel -- same test as _cross_sandwich_dev)
            sorted_k3 = (self._dev().data.numel() <= SORTED_K3_NNZ_PER_ROW * self.shape[0]
                         and self.shape[1] >= 1024)
            ent = self._ent() if dense_width > 64 and not sorted_k3 else None
            if dense_width <= 64 or (ent is None and self._lg() is None):
                self._ell(wide=dense_width > 64)
        if ent is None:
            self._slab()       # (categorical x sparse runs on the entry twin when there is one)
        # the twins are built: from here on the hot kernels read them, the 16-bit column twin (unrestricted matvec /
        # transpose_matvec) or nothing of the CSR columns at all -- the int32 column array goes (ext/_types.py)
        # (only where the unrestricted matvec kernels stream the 16-bit twin anyway: large blocks, TABMAT_AMD_CSR_U16)
        if xs.CSR_U16 and int(self._dev().data.numel()) >= xs.CSR_U16_MIN_NNZ:
            self._dev().compact_indices()
        return self

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def ndim(self):
        return 2

    @property
    def indices(self):
        return self._host().indices

    @property
    def indptr(self):
        return self._host().indptr

    @property
    def data(self):
        return self._host().data

    __array_ufunc__ = None

    def tocsc(self, copy=False):
        return self._host().tocsc(copy=copy)

    def unpack(self):
        return self._host()

    def toarray(self):
        return self._host().toarray()

    def dot(self, other):
        return self._host().dot(other)

    def transpose(self):
        return type(self)(self._host().T)

    T = property(transpose)

    def getcol(self, i):
        return type(self)(self._host()[:, [i]], column_names=[self._colnames[i]],
                          term_names=[self._terms[i]])

    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        return type(self)(self._host().astype(dtype, casting, copy))

    def __getitem__(self, key):
        row, col = check_indexer(key)
        if self._devblk is not None and selects_all_columns(col, self.shape[1]):
            # row indexing of the CSR twin in HBM (no host round trip)
            sub = type(self).from_device(self._devblk.take_rows(*device_row_index(row, self.shape[0])))
            sub._colnames, sub._terms = list(self._colnames), list(self._terms)
            return sub
        names = np.array(self._colnames, dtype=object)[col].ravel().tolist()
        terms = np.array(self._terms, dtype=object)[col].ravel().tolist()
        return type(self)(self._host()[row, col], column_names=names, term_names=terms)

    def __matmul__(self, other):
        return self.matvec(other)

    def multiply(self, other):
        if other.ndim == 1:
            return type(self)(self._host().multiply(other[:, np.newaxis]))
        return type(self)(self._host().multiply(other))

    # ---- hot path -----------------------------------------------------------------------
    def _row_parts(self):
        """None, or [(r0, r1, SparseMatrix)] row slices with fewer than PART_NNZ nonzeros each (the
        twins index entries with 32 bits)."""
        parts = self.__dict__.get("_parts_cache", False)
        if parts is False:
            parts = None
            ptr = self._dev().indptr
            nnz = int(self._dev().data.numel())
            if nnz >= PART_NNZ:
                n = self.shape[0]
                k = -(-nnz // max(1, int(0.8 * PART_NNZ)))
                targets = torch.arange(1, k, device=ptr.device, dtype=torch.int64) * nnz // k
                cuts = sorted(set([0] + torch.searchsorted(ptr, targets).clamp_(0, n).tolist() + [n]))
                parts = [(a, b, self[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            self.__dict__["_parts_cache"] = parts
        return parts

    def _narrow_pays(self, w: int) -> bool:
        """Host cost model (fitted on scripts/dev/time_sparse_cols.py, MI355X): the dense-block form
        of a w-column selection (zero + write + syrk of an n x w block, one atomic per selected
        entry) against the unrestricted self sandwich (per (row, tile) visit or per pair)."""
        n, mcols = self.shape
        nnz = float(self._dev().data.numel())
        per_row = nnz / max(n, 1)
        pairs = n * per_row * (per_row + 1) / 2
        nch = -(-mcols // 128)
        t_full = max(31e-12 * (nch * (nch + 1) / 2) * n, 1.6e-12 * pairs)
        if getattr(self, "_direct_pays", None):
            t_full = pairs / 22e9
        t_narrow = 0.2e-3 + n * w * 4.5e-12 + nnz * w / max(mcols, 1) * 40e-12
        return t_narrow < 0.8 * t_full

    def _sandwich_dev(self, d, rows, cols):
        parts = self._row_parts()
        if parts is not None:
            out = None
            for a, b, part in parts:
                r = None
                if rows is not None:
                    r64 = rows.to(torch.int64)
                    sel = r64[(r64 >= a) & (r64 < b)]
                    if sel.numel() == 0:
                        continue
                    r = (sel - a).to(torch.int32)
                res = part._sandwich_dev(d[a:b], r, cols)
                out = res if out is None else out + res
            if out is None:
                k = self.shape[1] if cols is None else D.nlen(cols)
                out = D.zeros((k, k), d.dtype)
            return out
        A = self._dev()
        from . import categorical_matrix as _cm

        if _cm.DETERMINISTIC and A.data.numel() > 0 and self.shape[0] > 0:
            res = self._sandwich_deterministic(d, rows, cols)
            if res is not None:
                return res
            import warnings

            # (2^28 rows or more in one block, or a stream of 2^27 batches: no entry twin at all)
            warnings.warn("TABMAT_AMD_DETERMINISTIC=1: this sparse block has no entry twin, its self sandwich "
                          "falls back to the LDS-atomic kernels and is NOT bit-reproducible from run to run",
                          RuntimeWarning, stacklevel=3)
        if getattr(self, "_direct_pays", None) is None:
            self._direct_pays = xs.direct_sandwich_pays(A)
        w = D.nlen(cols) if cols is not None else 0
        if (0 < w <= NARROW_COLS and 2 * w < self.shape[1] and self.shape[0] > 0 and A.data.numel() > 0
                and self._narrow_pays(w)
                and self.shape[0] * w * A.data.element_size() * 4 < torch.cuda.mem_get_info()[0]):
            # a narrow column selection (a solver's active set): the selected columns written out
            # as one row-major dense block from the CSC form (only their entries are read), then
            # the MFMA syrk -- cost in proportion to the selection (ext/sparse.pyx:17-77 with `cols`)
            from .ext import dense as xd
            from .ext._types import DenseDev

            rws, vls, bstart, _, col_bptr = A.csc_blocks()
            mx = getattr(self, "_max_col_len", None)
            if mx is None:
                ends = bstart[col_bptr]
                mx = self._max_col_len = int((ends[1:] - ends[:-1]).max().item())
            cd = cols.to(torch.int64)
            seg = torch.stack([bstart[col_bptr[cd]], bstart[col_bptr[cd + 1]]], dim=1).contiguous()
            T = torch.zeros((self.shape[0], w), dtype=A.data.dtype, device=A.data.device)
            xs.csc_densify_cols(rws, vls, seg, torch.arange(w, dtype=torch.int32, device=A.data.device),
                                mx, T)
            return xd.dense_sandwich(DenseDev(T, self.shape[0], w, 0), d, rows, None)
        pp = getattr(self, "_pairs_pay", None)
        if pp is None:
            pp = self._pairs_pay = xs.pairs_sandwich_pays(A)
        if pp and (rows is None or D.nlen(rows) > 0.3 * self.shape[0]):
            # wide block: the pair-stream kernel (cost follows the entries and pairs of a row, csrc/sparse_pairs.hip);
            # a long row list is a masked d (the kernel never touches the entries of a row with d == 0); short
            # ones keep the row-list kernels below
            d = D.masked_d(d, rows, as_set=True)    # (ext/sparse.pyx:46-48: a row mask)
            res = xs.sparse_sandwich_pairs(A, d)
            if cols is not None:
                c64 = cols.to(torch.int64)
                res = res[c64][:, c64].contiguous()
            return res
        pays = getattr(self, "_direct_pays", None)
        if pays is None:
            pays = self._direct_pays = xs.direct_sandwich_pays(A)
        if pays:
            # wide and very sparse: cost per pair instead of per (row, tile)
            d = D.masked_d(d, rows, as_set=True)    # (ext/sparse.pyx:46-48: a row mask)
            res = xs.sparse_sandwich_direct(A, d)
            if cols is not None:
                c64 = cols.to(torch.int64)
                res = res[c64][:, c64].contiguous()
            return res
        if A.data.numel() > 0 and A.data.numel() < 2**31 and self.shape[1] <= 128 * 32:
            if rows is not None and 0 < D.nlen(rows) <= ROW_LIST_FRACTION * self.shape[0]:
                # short row list: the same pipeline over the selected rows only
                res = xs.sparse_sandwich_rows(A, d, rows)
                if cols is not None:
                    c64 = cols.to(torch.int64)
                    res = res[c64][:, c64].contiguous()
                return res
            # fast path: unrestricted chunk-pointer kernel; row restriction = masked d,
            # column restriction = sub-selection of the small result
            d = D.masked_d(d, rows, as_set=True)    # (ext/sparse.pyx:46-48: a row mask)
            bl = getattr(self, "_blocks_pay", None)
            if bl is None:
                bl = self._blocks_pay = xs.blocks_sandwich_pays(A)
            res = xs.sparse_sandwich_blocks(A, d) if bl else xs.sparse_sandwich_chunked(A, d)
            if cols is not None:
                c64 = cols.to(torch.int64)
                res = res[c64][:, c64].contiguous()
            return res
        return xs.sparse_sandwich(A, d, rows, cols)

    def _ent_det(self):
        """The entry twin for the deterministic mode: the one the products use, else (a block so sparse that the
        padded stream exceeds ELL_MAX_PAD x its nonzeros) one built without that limit -- reproducibility was
        asked for, the padding is its price."""
        ent = self._ent()
        if ent is None:
            hit = getattr(self, "_entblk_det", None)
            if hit is None:
                twin = SlabEnt.from_csr(self._dev(), max_pad=None)
                hit = self._entblk_det = twin if twin is not None else False
            ent = hit or None
        return ent

    def _sandwich_deterministic(self, d, rows, cols=None):
        """TABMAT_AMD_DETERMINISTIC=1: the sparse self sandwich with a FIXED summation order, bit-identical
        from run to run -- the property the reference's kernel has by construction (thread-owned output rows,
        ext/sparse.pyx:55-74) and the LDS-atomic pair kernels (K2 / K2b) do not.  The block's columns are
        written out densely 128 at a time and every chunk goes through the entry-list kernel (K3,
        csrc/sparse_ent.hip): an accumulator there receives its entries in stream order, the workgroups'
        partial sums are added in a fixed order.  out[i, j] and out[j, i] see the same terms in the same
        order but multiply them in a different one, so the lower triangle is mirrored.  About 4x the time of
        the atomic kernel at BASELINE configs[3]; opt-in.  With `cols` only the selected columns are written out
        (k / 128 passes instead of m / 128) and the (k, k) result of the selection is returned."""
        from .ext._types import DenseDev

        ent = self._ent_det()
        if ent is None:
            return None
        n, m = self.shape
        A = self._dev()
        d = D.masked_d(d, rows, as_set=True)    # (ext/sparse.pyx:46-48: a row mask)
        inv = None
        if cols is None:
            sel = torch.arange(m, dtype=torch.int64, device=d.device)
        else:
            # a column id may occur twice (normalize_index allows it, as the reference does): the product is
            # computed over the DISTINCT columns -- `colmap` below holds one slot per column id -- and the
            # repeats are expanded from the small result
            sel, inv = torch.unique(cols.to(torch.int64), return_inverse=True)
            if int(sel.numel()) == int(inv.numel()) and bool((sel == cols.to(torch.int64)).all()):
                inv = None
        k = int(sel.numel())
        out = torch.empty((m, k), dtype=d.dtype, device=d.device)
        W = 128
        for c0 in range(0, k, W):
            w = min(W, k - c0)
            w_pad = (w + 3) // 4 * 4                     # 16-byte aligned rows for the kernel's slab copy
            colmap = torch.full((m,), -1, dtype=torch.int32, device=d.device)
            colmap[sel[c0:c0 + w]] = torch.arange(w, dtype=torch.int32, device=d.device)
            T = torch.zeros((n, w_pad), dtype=A.data.dtype, device=d.device)
            xs.csr_densify_cols(A, colmap, T)
            res = xs.csr_dense_sandwich_ent(ent, DenseDev(T, n, w_pad, 0), d)
            out[:, c0:c0 + w] = res[:, :w]
            del T
        out = out if cols is None else out[sel]
        low = torch.tril(out)
        full = low + torch.tril(out, -1).T
        if inv is not None:
            full = full.index_select(0, inv).index_select(1, inv)
        return full

    def sandwich(self, d, rows=None, cols=None):
        """sparse_matrix.py:175-185."""
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        res = self._sandwich_dev(D.to_dev(d), D.idx_dev(normalize_index(rows, self.shape[0])),
                                 D.idx_dev(normalize_index(cols, self.shape[1])))
        return res if on_dev else D.to_host(res)

    def _cross_sandwich_dev(self, other, d, rows, L_cols, R_cols, colsum_box=None):
        """colsum_box: a list that receives self[rows, L_cols]' d[rows] when the kernel that runs
        produces it in the same pass (the lane-group K3; SplitMatrix._sandwich_xtd_dev)."""
        from .categorical_matrix import CategoricalMatrix
        from .dense_matrix import DenseMatrix

        if isinstance(other, DenseMatrix):
            if other.dtype != self.dtype:
                raise TypeError(
                    "self, B and d all need to be of same dtype, either np.float64 or "
                    f"np.float32. This matrix is of type {self.dtype}, B is of type "
                    f"{other.dtype}.")
            Bd = other._dev_c()
            A = self._dev()
            if (rows is not None and 0 < D.nlen(rows) <= ROW_LIST_FRACTION_K3 * self.shape[0]
                    and 0 < A.data.numel() < 2**31 and self.shape[1] <= 128 * 32):
                # short row list: row-list kernel on the chunk-major twin (cost ~ len(rows))
                res = xs.csr_dense_sandwich_rows(A, Bd, d, rows)
                if L_cols is not None:
                    res = res[L_cols.to(torch.int64)]
                if R_cols is not None:
                    res = res[:, R_cols.to(torch.int64)]
                return res
            if self.shape[0] > 0 and self._dev().data.numel() > 0 and (
                    rows is None or self._values_finite()):
                # fast path: unrestricted slab kernel; a row restriction is a masked d (excluded
                # rows contribute exactly 0), column restrictions select from the small result
                d = D.masked_d(d, rows)
                if (A.data.numel() <= SORTED_K3_NNZ_PER_ROW * self.shape[0] and self.shape[1] >= 1024
                        and xs.ell_supported(Bd)):      # (to_device applies the same test before building twins)
                    # a few nonzeros per row in a wide block: column by column on the CSC form
                    res = xs.csc_dense_sandwich_sorted(A, Bd, d)
                    if L_cols is not None:
                        res = res[L_cols.to(torch.int64)]
                    if R_cols is not None:
                        res = res[:, R_cols.to(torch.int64)]
                    return res
                wide = Bd.m > 64 and xs.ell_supported(Bd)
                ent = self._ent() if wide else None
                lg = self._lg() if (wide and ent is None) else None
                ell = None
                if ent is None and lg is None and xs.ell_supported(Bd):
                    ell = self._ell(wide=Bd.m > 64)
                if ent is not None and colsum_box is not None:
                    res, cs = xs.csr_dense_sandwich_ent(ent, Bd, d, want_colsum=True)
                    colsum_box.append(cs if L_cols is None else cs[L_cols.to(torch.int64)])
                elif ent is not None:
                    res = xs.csr_dense_sandwich_ent(ent, Bd, d)
                elif lg is not None and colsum_box is not None:
                    res, cs = xs.csr_dense_sandwich_lg(lg, Bd, d, want_colsum=True)
                    colsum_box.append(cs if L_cols is None else cs[L_cols.to(torch.int64)])
                elif lg is not None:
                    res = xs.csr_dense_sandwich_lg(lg, Bd, d)
                elif ell is not None:
                    res = xs.csr_dense_sandwich_ell(ell, Bd, d)
                else:
                    res = xs.csr_dense_sandwich_slab(self._slab(), Bd, d)
                if L_cols is not None:
                    res = res[L_cols.to(torch.int64)]
                if R_cols is not None:
                    res = res[:, R_cols.to(torch.int64)]
                return res
            return xs.csr_dense_sandwich(self._dev(), Bd, d, rows, L_cols, R_cols)
        if isinstance(other, CategoricalMatrix):
            return other._cross_sandwich_dev(self, d, rows, R_cols, L_cols).T
        raise TypeError

    def _cross_sandwich(self, other, d, rows, L_cols=None, R_cols=None):
        """sparse_matrix.py:187-204."""
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        check_sandwich_compatible(self, d)  # dtype rule of sparse_matrix.py:218-223
        res = self._cross_sandwich_dev(
            other, D.to_dev(d), D.idx_dev(normalize_index(rows, self.shape[0])),
            D.idx_dev(normalize_index(L_cols, self.shape[1])),
            D.idx_dev(normalize_index(R_cols, other.shape[1])))
        return res if on_dev else D.to_host(res)

    def sandwich_dense(self, B, d, rows, L_cols, R_cols):
        """self.T @ diag(d) @ B for a dense ndarray / DenseMatrix B (sparse_matrix.py:206-229)."""
        from .dense_matrix import DenseMatrix

        Bm = B if isinstance(B, DenseMatrix) else DenseMatrix(B)
        return self._cross_sandwich(Bm, d, rows, L_cols, R_cols)

    def _matvec_dev(self, vec, rows, cols, out, transpose):
        fn = xs.csc_rmatvec if transpose else xs.csr_matvec
        return fn(self._dev(), vec, rows, cols, out)

    def _matvec_helper(self, vec, rows, cols, out, transpose):
        on_dev = D.is_dev(vec)
        if not on_dev:
            vec = np.asarray(vec)
        check_matvec_dimensions(self, vec, transpose)
        n, m = self.shape
        rows_n = normalize_index(rows, n)
        cols_n = normalize_index(cols, m)
        if rows_n is not None and len(rows_n) == n:
            rows_n = None
        if cols_n is not None and len(cols_n) == m:
            cols_n = None
        tdt = D.torch_dtype(self.dtype)
        v_dev = D.to_dev(vec, tdt)
        rd, cd = D.idx_dev(rows_n), D.idx_dev(cols_n)
        if v_dev.ndim == 1:
            res = self._matvec_dev(v_dev, rd, cd, None, transpose)
        else:
            res = xs.csr_matvec_multi(self._dev(), v_dev, rd, cd, transpose)
        if not on_dev:
            res = D.to_host(res)
        if out is None:
            return res
        if transpose and cols_n is not None:
            out[cols_n if not D.is_dev(out) else D.idx_dev(cols_n, torch.int64)] += res
        else:
            out += res
        return out

    def matvec(self, vec, cols=None, out=None):
        """sparse_matrix.py:277-282."""
        check_matvec_out_shape(self, out)
        return self._matvec_helper(vec, None, cols, out, False)

    def transpose_matvec(self, vec, rows=None, cols=None, out=None):
        """sparse_matrix.py:284-293."""
        check_transpose_matvec_out_shape(self, out)
        return self._matvec_helper(vec, rows, cols, out, True)

    def _get_col_stds(self, weights, col_means):
        """sparse_matrix.py:295-311 with the K7 kernel (ext/sparse.pyx:262-282)."""
        tdt = D.torch_dtype(self.dtype)
        ex2 = D.to_host(xs.transpose_square_dot_weights(self._dev(),
                                                        D.to_dev(np.asarray(weights), tdt)))
        arg = ex2 - np.asarray(col_means) ** 2
        arg[arg < 0] = 0
        return np.sqrt(arg)
