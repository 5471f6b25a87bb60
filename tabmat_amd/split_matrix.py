"""SplitMatrix: column-wise union of one dense, one sparse and any number of categorical
blocks (reference: /root/reference/src/tabmat/split_matrix.py).  The cross-block assembly of
sandwich / matvec / transpose_matvec runs on ONE GPU: every block product is a HIP kernel and
the block results are scattered into the float64 p x p (or length-p / length-n) device buffer
by tm_scatter_block -- no host round trip inside a call."""
from __future__ import annotations

import os
import warnings
from collections.abc import Sequence
from typing import Optional, Union

import numpy as np
import torch
from scipy import sparse as sps

from . import _device as D
from .categorical_matrix import CategoricalMatrix
from .dense_matrix import DenseMatrix
from .ext import split as xsplit
from .matrix_base import MatrixBase
from .sparse_matrix import SparseMatrix
from .standardized_mat import StandardizedMatrix
from .util import (
    check_matvec_dimensions,
    check_matvec_out_shape,
    check_sandwich_compatible,
    check_transpose_matvec_out_shape,
    collapse_identity,
    normalize_index,
    set_up_rows_or_cols,
)


# (Rounds 2 and 3 measured two ways of running the block products of one sandwich concurrently -- k HIP streams,
# and the dense syrk as a co-resident guest kernel -- at 0.0-0.3 ms gain for a 15 ms step: every big kernel fills
# a compute unit's LDS or registers, and the f64 syrk queues in the same VALU issue slots as its partners
# (profiles/r2_microbench.txt, profiles/r3_coresidency.txt).  The harnesses live in scripts/dev/overlap_step.py;
# the product path launches its kernels in line on the current stream.)
# row lists shorter than this share of n take the row-list form of the fused categorical x sparse term
ROW_LIST_FRACTION_CATSPARSE = 0.04      # (measured break-even, profiles/r3_cols_rows.txt)
# a categorical block's diagonal as the row sum of its table with a complete partner categorical
DIAG_FROM_PAIRS = True
# all small categorical x categorical tables + diagonals in one launch (tm_multi_cat_pairs_*)
CAT_PAIRS_FUSED = os.environ.get("TABMAT_AMD_CAT_PAIRS", "1") != "0"
# a column selection that keeps at least this share of the columns is computed as the UNRESTRICTED
# product followed by a selection of the result (every entry of X'DX / X'v depends on its own
# columns only, so the entries are the same; the unrestricted kernels are the tuned ones).  matvec:
# zeros in the coefficient vector instead.
FULL_THEN_SELECT = float(os.environ.get("TABMAT_AMD_FULL_THEN_SELECT", "0.5"))
# ... but only while the full (p, p) result stays small and the blocks WITHOUT a selected column (skipped by the
# restricted path, as by the reference) are at most this share of the product's work (_full_product_pays)
FULL_RESULT_MAX_BYTES = 256 << 20
FULL_UNSELECTED_SHARE = 0.2
# matvec / transpose_matvec stream all of X whatever the selection (row-major dense rows, CSR): the
# unrestricted kernels are never slower (cfg4 shape, 2M rows, 5 % of the columns: 0.88 / 0.99 ms
# restricted, 0.55 / 0.60 ms unrestricted + selection), so they always take this form
FULL_THEN_SELECT_MV = 0.0
# a NARROW selection (below FULL_THEN_SELECT, at most NARROW_COLS selected dense + sparse columns -- one
# syrk panel):
# those columns are gathered / written out densely into one row-major block and the unrestricted
# kernels run on [that block | the categorical blocks] -- cost in proportion to the selection, as
# in the reference, instead of the full K2 / K3 passes of the generic restricted kernels
NARROW_COLS = int(os.environ.get("TABMAT_AMD_NARROW_COLS", "128"))
# Entry indices inside one sparse block's twins are 32-bit: a SplitMatrix whose sparse block holds
# this many nonzeros or more is worked on in ROW PARTS (the sandwich is a sum over rows), each with
# twins of its own -- 288 GB of HBM hold blocks of several 10^9 nonzeros.
from . import categorical_matrix as _cm
from . import sparse_matrix as _spm      # PART_NNZ lives there (also used by SparseMatrix itself)


class _Centering:
    """Column centres of a CENTRED sandwich (StandardizedMatrix.sandwich, standardized_mat.py:123-172): vec[b] =
    device tensor over ALL columns of dense block b (block dtype).  The self term of such a block is computed
    as (X_b - 1 c')' D (X_b - 1 c') by the dense kernels (the centre is subtracted on the way in); every other
    block product stays raw and is centred afterwards with rank-one terms (tm_standardize_sandwich_centered_f64).
    cs_centered collects the blocks whose column sums came out of a centred kernel ((X_b - 1 c')' d).
    groups: None = the entries computed centred are exactly the self terms of the blocks in vec; the narrow
    column-selection path (which computes dense AND sparse columns as one dense block, so their cross entries
    come out centred too) leaves an int32 device vector over the RESULT's columns here instead: entries (i, j)
    with groups[i] == groups[j] >= 0 are centred, all others raw."""
    NARROW_GROUP = 1 << 20

    def __init__(self, vec):
        self.vec = dict(vec)
        self.cs_centered = set()
        self.groups = None

    def get(self, b):
        return self.vec.get(b)


def as_tabmat(a):
    """split_matrix.py:22-37."""
    if isinstance(a, (MatrixBase, StandardizedMatrix)):
        return a
    if sps.issparse(a):
        return SparseMatrix(a.tocsc(copy=False))
    if isinstance(a, np.ndarray):
        return DenseMatrix(a)
    raise ValueError(f"Cannot convert type {type(a)} to Matrix.")


def hstack(tup: Sequence) -> MatrixBase:
    """split_matrix.py:40-62."""
    mats = [as_tabmat(a) for a in tup]
    if not mats:
        raise ValueError("Need at least one array to concatenate.")
    if all(isinstance(m, SparseMatrix) for m in mats):
        return SparseMatrix(sps.hstack([m.array_csc for m in mats]))
    if all(isinstance(m, DenseMatrix) for m in mats):
        return DenseMatrix(np.hstack([m.toarray() for m in mats]))
    return SplitMatrix(mats)


def _merge_same_kind(matrices, indices):
    """All dense blocks -> one DenseMatrix, all sparse blocks -> one SparseMatrix, columns
    ordered by global index; categoricals untouched (split_matrix.py:85-141)."""
    for kind in (DenseMatrix, SparseMatrix):
        which = [i for i, m in enumerate(matrices) if isinstance(m, kind)]
        if len(which) <= 1:
            continue
        gidx = np.concatenate([indices[i] for i in which])
        order = np.argsort(gidx)
        names = np.concatenate([np.array(matrices[i]._colnames, dtype=object) for i in which])
        terms = np.concatenate([np.array(matrices[i]._terms, dtype=object) for i in which])
        if kind is DenseMatrix:
            merged = DenseMatrix(np.hstack([matrices[i].toarray() for i in which])[:, order])
        else:
            merged = SparseMatrix(sps.hstack([matrices[i].array_csc for i in which]).tocsc()[:, order])
        merged._colnames = names[order].tolist()
        merged._terms = terms[order].tolist()
        matrices[which[0]] = merged
        indices[which[0]] = gidx[order]
        drop = set(which[1:])
        matrices = [m for i, m in enumerate(matrices) if i not in drop]
        indices = [x for i, x in enumerate(indices) if i not in drop]
    return matrices, indices


class SplitMatrix(MatrixBase):
    """matrices: the blocks; indices: for each block the (sorted) global columns it covers."""

    def __init__(self, matrices: Sequence[MatrixBase], indices: Optional[list] = None):
        flat, corrections = [], []
        for mat in matrices:
            if not isinstance(mat, MatrixBase):
                raise ValueError(
                    "Expected all elements of matrices to be subclasses of MatrixBase.")
            if isinstance(mat, SplitMatrix):
                offset = 0
                for iind, imat in zip(mat.indices, mat.matrices):
                    flat.append(imat)
                    corrections.append(iind - np.arange(len(iind), dtype=np.int64) - offset)
                    offset += len(iind)
            else:
                flat.append(mat)
                corrections.append(np.zeros(mat.shape[1], dtype=np.int64))
        self.dtype = flat[0].dtype
        n_row = flat[0].shape[0]
        for i, mat in enumerate(flat):
            if mat.dtype != self.dtype:
                warnings.warn("Matrices do not all have the same dtype. Dtypes are "
                              f"{[elt.dtype for elt in flat]}.")
            if mat.shape[0] != n_row:
                raise ValueError(
                    "All matrices should have the same first dimension, "
                    f"but the first matrix has first dimension {n_row} and matrix {i} "
                    f"has first dimension {mat.shape[0]}.")
        if indices is None:
            indices, cur = [], 0
            for mat, corr in zip(flat, corrections):
                indices.append(np.arange(cur, cur + mat.shape[1], dtype=np.int64) + corr)
                cur += mat.shape[1]
            n_col = cur
        else:
            indices = [np.asarray(ix) for ix in indices]
            allidx = np.concatenate(indices) if indices else np.zeros(0, dtype=np.int64)
            n_col = len(allidx)
            if (np.arange(n_col, dtype=np.int64) != np.sort(allidx)).any():
                raise ValueError("Indices should contain all integers from 0 to one less than "
                                 "the number of columns.")
            for i, ix in enumerate(indices):
                if not xsplit.is_sorted(ix):
                    raise ValueError(
                        f"Each index block should be sorted, but indices[{i}] was not sorted")
        for i, (mat, ix) in enumerate(zip(flat, indices)):
            if mat.shape[1] != len(ix):
                raise ValueError(
                    f"Element {i} of indices should should have length {mat.shape[1]}, "
                    f"but it has shape {np.asarray(ix).shape}")
        keep = [i for i, m in enumerate(flat) if m.shape[1] > 0]
        mats, idxs = _merge_same_kind([flat[i] for i in keep], [indices[i] for i in keep])
        self.matrices = mats
        self.indices = [np.asarray(ix, dtype=np.int64) for ix in idxs]
        self.shape = (n_row, n_col)
        self._dev_indices = None
        assert self.shape[1] > 0

    # ---- bookkeeping ----------------------------------------------------------------------
    def _split_col_subsets(self, cols):
        """split_matrix.py:269-291: (positions in the result, local columns per block, n_cols)."""
        if cols is None:
            return self.indices, [None] * len(self.indices), self.shape[1]
        return xsplit.split_col_subsets(self, set_up_rows_or_cols(cols, self.shape[1]))

    def _onehot_slab(self, cat_ids):
        key = tuple(cat_ids)
        cache = self.__dict__.setdefault("_onehot_cache", {})
        if key not in cache:
            from .ext._types import onehot_slab

            mats = self.matrices
            cats = [(mats[i]._dev(), mats[i].shape[1], mats[i].drop_first) for i in cat_ids]
            cache[key] = onehot_slab(cats, self.shape[0], D.torch_dtype(self.dtype))
        return cache[key]

    def _dev_idx(self, arrs):
        return [D.idx_dev(a, torch.int64) for a in arrs]

    def _full_dev_indices(self):
        if self._dev_indices is None:
            self._dev_indices = self._dev_idx(self.indices)
        return self._dev_indices

    def to_device(self):
        parts = self._parts()
        if parts is not None:
            for _, _, p in parts:
                p.to_device()
            self._full_dev_indices()
            return self
        dense_w = [m.shape[1] for m in self.matrices if isinstance(m, DenseMatrix)]
        for m in self.matrices:
            if isinstance(m, SparseMatrix):
                m.to_device(dense_width=dense_w[0] if dense_w else None)
            else:
                m.to_device()
        self._full_dev_indices()
        cat_ids = [i for i, m in enumerate(self.matrices)
                   if isinstance(m, CategoricalMatrix) and m.shape[1] > 0]
        for grp in self._cat_groups(cat_ids):
            cats = [(self.matrices[i]._dev(), self.matrices[i].shape[1], self.matrices[i].drop_first)
                    for i in grp]
            if any(isinstance(m, DenseMatrix) and not xsplit.multi_cat_dense_wide_ok(cats, m._dev_c())
                   and not xsplit.multi_cat_dense_tile_ok(cats, m._dev_c())
                   and not xsplit.cat_dense_sorted_ok(m._dev_c()) for m in self.matrices):
                self._onehot_slab(grp)
        others = any(not isinstance(m, CategoricalMatrix) for m in self.matrices)
        for i in cat_ids:            # many levels: the row grouping of the level-sorted kernels
            if self.matrices[i].shape[1] > self.FUSED_LEVELS and others:
                self.matrices[i]._det_plan()
        return self

    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        return SplitMatrix([m.astype(dtype=dtype, order=order, casting=casting, copy=copy)
                            for m in self.matrices], self.indices)

    def toarray(self) -> np.ndarray:
        out = np.empty(self.shape)
        for mat, idx in zip(self.matrices, self.indices):
            out[:, idx] = mat.toarray()
        return out

    def getcol(self, i: int):
        i %= self.shape[1]
        for mat, idx in zip(self.matrices, self.indices):
            loc = np.where(idx == i)[0]
            if len(loc):
                return mat.getcol(int(loc[0]))
        raise RuntimeError(f"Column {i} was not found.")

    def __getitem__(self, key):
        if isinstance(key, tuple):
            row, col = key
        else:
            row, col = key, slice(None)
        if not (isinstance(col, slice) and col == slice(None)):
            raise NotImplementedError(f"Only row indexing is supported. Index passed was {key}.")
        if isinstance(row, int):
            row = [row]
        return SplitMatrix([m[row, :] for m in self.matrices], self.indices)

    def __repr__(self):
        return "SplitMatrix(" + ", ".join(type(m).__name__ + str(m.shape) for m in self.matrices) + ")"

    def multiply(self, other):
        return SplitMatrix([m.multiply(other) for m in self.matrices], self.indices)

    def get_names(self, type="column", missing_prefix=None, indices=None):
        names = np.empty(self.shape[1], dtype=object)
        for idx, mat in zip(self.indices, self.matrices):
            names[idx] = mat.get_names(type, missing_prefix, idx)
        return list(names)

    def set_names(self, names, type="column"):
        names = np.asarray([names] if isinstance(names, str) else names, dtype=object)
        if len(names) != self.shape[1]:
            raise ValueError(f"Length of names must be {self.shape[1]}")
        for idx, mat in zip(self.indices, self.matrices):
            mat.set_names(names[idx].tolist(), type)

    def _get_col_means(self, weights):
        means = np.empty(self.shape[1], dtype=self.dtype)
        for idx, mat in zip(self.indices, self.matrices):
            means[idx] = mat._get_col_means(weights)
        return means

    def _get_col_stds(self, weights, col_means):
        stds = np.empty(self.shape[1], dtype=self.dtype)
        for idx, mat in zip(self.indices, self.matrices):
            stds[idx] = mat._get_col_stds(weights, col_means[idx])
        return stds

    # ---- hot path -------------------------------------------------------------------------
    def _sandwich_plan(self, cols_host):
        """Column bookkeeping of one sandwich call, staged on the device once: output positions
        and per-block column subsets (split_matrix.py:341-349)."""
        if cols_host is None:
            return self._full_dev_indices(), [None] * len(self.indices), self.shape[1]
        # the last selection's index arrays stay on the device: a solver repeats its active set,
        # and the 2 uploads per block were 0.8 ms for 20 blocks
        key = np.asarray(cols_host).tobytes()
        hit = self.__dict__.get("_plan_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        pos, sub_cols, n_cols = self._split_col_subsets(cols_host)
        pos_d = self._dev_idx(pos)
        sub_d = [None if sc is None else D.idx_dev(sc) for sc in sub_cols]
        self._cols_dev64(cols_host)
        if CAT_PAIRS_FUSED:
            plan = self._cat_pairs_plan()
            if plan is not None and plan.n_pairs > 0:
                self._pairs_pos_sel(plan, cols_host, (pos, sub_cols))    # staged here (no copies
        self.__dict__["_plan_cache"] = (key, (pos_d, sub_d, n_cols))    # inside a graph capture)
        return pos_d, sub_d, n_cols

    def _blocks_finite(self) -> bool:
        """No stored inf / nan in any dense or sparse block (checked once per block)."""
        return all(mb._values_finite() for mb in self.matrices
                   if isinstance(mb, (DenseMatrix, SparseMatrix)))

    def _cols_dev64(self, cols_host):
        """The column selection as an int64 device tensor (last selection cached)."""
        key = np.asarray(cols_host).tobytes()
        hit = self.__dict__.get("_cols64")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_cols64"] = (key, D.idx_dev(cols_host, torch.int64))
        return hit[1]

    def _pairs_pos_sel(self, plan, cols_host, split=None):
        """Output position of every level of the plan's categoricals under the column selection
        `cols_host` (-1: not selected), device int64; the last selection is kept."""
        key = np.asarray(cols_host).tobytes()
        hit = self.__dict__.get("_pos_sel")
        if hit is not None and hit[0] == key:
            return hit[1]
        pos_h, sub_h = split if split is not None else self._split_col_subsets(cols_host)[:2]
        pf = np.full(int(plan.pstart[-1]), -1, dtype=np.int64)
        for a, i in enumerate(plan.cat_ids):
            if len(sub_h[i]):
                pf[int(plan.pstart[a]) + np.asarray(sub_h[i], dtype=np.int64)] = pos_h[i]
        t = D.to_dev(pf)
        self.__dict__["_pos_sel"] = (key, t)
        return t

    def _sandwich_xtd_dev(self, d, rows, cols_host, center=None):
        """(X' diag(d) X, X' d) restricted to rows / cols, both float64 on the device, from ONE
        pass over the blocks where the algebra allows it (what StandardizedMatrix.sandwich needs,
        standardized_mat.py:148-150, which makes two passes): the column sums of a categorical
        block are the diagonal of its own sandwich, and for any other block B they are the sums
        over the levels of its cross term with a COMPLETE categorical C (no dropped level, no
        missing codes: every row of C has exactly one 1, so C 1 = 1 and B' D 1 = (B' D C) 1).
        Blocks without such a partner get their own transpose_matvec launch.
        center (_Centering or None): the dense blocks it names enter their self term centred; two more results
        then: a bool device vector, True where xtd holds the CENTRED column sum (X - 1 c')' d, and the groups
        vector of _Centering (None: the default -- the centred entries are the named blocks' self terms)."""
        colsum = [None] * len(self.matrices)
        out = self._sandwich_dev(d, rows, cols_host, colsum=colsum, center=center)
        pos_d, sub_d, n_cols = self._sandwich_plan(cols_host)
        xtd = D.zeros((n_cols,), torch.float64)
        cmask = None if center is None else torch.zeros((n_cols,), dtype=torch.bool, device=xtd.device)
        for i, (mi, pd, sd) in enumerate(zip(self.matrices, pos_d, sub_d)):
            if sd is not None and D.nlen(sd) == 0:
                continue
            cs = colsum[i]
            if cs is None:
                if isinstance(mi, CategoricalMatrix):
                    full = D.zeros((mi.shape[1],), d.dtype)
                    mi._transpose_matvec_dev(d, rows, sd, full)
                    cs = full if sd is None else full[sd.to(torch.int64)]
                else:
                    cs = mi._matvec_dev(d, rows, sd, None, True)
            elif center is not None and i in center.cs_centered:
                cmask[pd] = True
            xtd[pd] = cs.to(torch.float64)
        if center is not None:
            return out, xtd, cmask, center.groups
        return out, xtd

    # levels that fit one LDS tile of doubles next to 32 dense / 33 sparse columns
    FUSED_LEVELS = 496
    FUSED_CATS = 8

    def _cat_groups(self, cat_ids):
        """Categorical blocks whose cross terms are fused into one pass (tm_multi_cat_*): groups of
        at most FUSED_CATS (8) blocks and FUSED_LEVELS stacked levels, in block order.  A categorical with more
        levels than that is left out: its cross terms go through the level-sorted kernels, whose
        cost does not depend on the number of levels (CategoricalMatrix._cross_sandwich_dev)."""
        groups, cur, tot = [], [], 0
        for i in cat_ids:
            k = self.matrices[i].shape[1]
            if k > self.FUSED_LEVELS:
                continue
            if cur and (len(cur) == self.FUSED_CATS or tot + k > self.FUSED_LEVELS):
                groups.append(cur)
                cur, tot = [], 0
            cur.append(i)
            tot += k
        if cur:
            groups.append(cur)
        return groups

    def _cat_pairs_plan(self):
        """Bundling of the categorical x categorical tables for tm_multi_cat_pairs_* (None when the
        matrix has no categorical block).  Static: built once."""
        plan = self.__dict__.get("_cp_plan", False)
        if plan is False:
            plan = None
            ids = [i for i, m in enumerate(self.matrices)
                   if isinstance(m, CategoricalMatrix) and m.shape[1] > 0]
            if len(ids) >= 1:
                pos = self._full_dev_indices()
                plan = xsplit.CatPairsPlan([(i, self.matrices[i].shape[1]) for i in ids],
                                           [pos[i] for i in ids])
            self.__dict__["_cp_plan"] = plan
        return plan

    def _cat_hist_plan(self):
        """All categorical blocks' weighted histograms in one launch (transpose_matvec): the
        diagonals-only form of the pair-table plan.  None with fewer than two categoricals."""
        plan = self.__dict__.get("_ch_plan", False)
        if plan is False:
            plan = None
            ids = [i for i, m in enumerate(self.matrices)
                   if isinstance(m, CategoricalMatrix) and m.shape[1] > 0 and m.shape[0] > 0]
            if len(ids) >= 2:
                pos = self._full_dev_indices()
                plan = xsplit.CatPairsPlan([(i, self.matrices[i].shape[1]) for i in ids],
                                           [pos[i] for i in ids], diag_only=True)
            self.__dict__["_ch_plan"] = plan
        return plan

    def _fused_cats(self, mw, cats, cat_ids, d_eff, rows, total, budget, d_rows=None):
        """All categorical x `mw` cross blocks from ONE pass over `mw` (tm_multi_cat_*), stacked
        [sum of levels, mw columns], or None when no fused kernel applies."""
        if isinstance(mw, DenseMatrix) and xsplit.multi_cat_dense_wide_ok(cats, mw._dev_c()):
            # few enough levels for one LDS tile: one pass of 16-byte loads over the dense
            # block, one LDS atomic per (row, categorical, 16 columns)
            if rows is not None and D.nlen(rows) <= 0.5 * self.shape[0]:
                # short row list: only those rows of the dense block are read
                return xsplit.multi_cat_dense_sandwich(cats, d_rows, mw._dev_c(), rows)
            return xsplit.multi_cat_dense_sandwich(cats, d_eff, mw._dev_c())
        if isinstance(mw, DenseMatrix) and xsplit.multi_cat_dense_tile_ok(cats, mw._dev_c()):
            # a narrow dense block (any order / alignment): the generic LDS-tile kernel
            return xsplit.multi_cat_dense_sandwich(cats, d_eff, mw._dev_c())
        if isinstance(mw, DenseMatrix):
            if xsplit.cat_dense_sorted_ok(mw._dev_c()):
                return None          # pair by pair on the level-sorted kernel
            # (F-ordered / unaligned operand) stacked one-hot encodings as a
            # 1-nonzero-per-row-and-categorical sparse block in slab form -> atomic-free gather
            # kernel (sparse.hip, K3 v2)
            from .ext import sparse as xs

            oh, inv = self._onehot_slab(cat_ids)
            return xs.csr_dense_sandwich_slab(oh, mw._dev_c(), d_eff)[inv]
        if (isinstance(mw, SparseMatrix) and total * 33 <= budget and rows is not None
                and 0 < D.nlen(rows) <= ROW_LIST_FRACTION_CATSPARSE * self.shape[0]
                and 0 < mw._dev().data.numel() < 2**31):
            # short row list: only the selected rows' entries are read (the row kernel walks
            # dependent loads per row: 0.35 ms at 10 % of 2M rows against 0.19 ms for the masked
            # full pass, which it beats below ~5 %, profiles/r3_cols_rows.txt)
            return xsplit.multi_cat_sparse_sandwich_rows(cats, d_rows, mw._dev(), rows)
        if (isinstance(mw, SparseMatrix) and total * 33 <= budget
                and mw._dev().data.numel() > 0 and (rows is None or mw._values_finite())):
            # on the entry twin when the sparse x dense term of this matrix runs on it anyway (a C-ordered dense
            # block of more than 64 columns): no slab-form twin is built for the block at all
            ent = None
            if len(cats) <= 8 and (getattr(mw, "_entblk", None) or any(
                    isinstance(mo, DenseMatrix) and mo.shape[1] > 64 and mo.dtype == mw.dtype
                    for mo in self.matrices)):
                ent = mw._ent()
            if ent is not None:
                pk_cache = self.__dict__.setdefault("_packed_codes", {})
                key = tuple(cat_ids)
                if key not in pk_cache:          # (static per set of categoricals: one word of tile rows per row)
                    pk_cache[key] = xsplit.pack_codes(cats)
                return xsplit.multi_cat_sparse_sandwich_ent(cats, d_eff, ent, pk_cache[key])
            return xsplit.multi_cat_sparse_sandwich(cats, d_eff, mw._slab())
        return None

    def _parts(self):
        """None, or [(r0, r1, SplitMatrix over rows r0 .. r1 - 1)] when a sparse block holds PART_NNZ
        nonzeros or more.  The parts are device row slices (views of the block storage); their twins
        are built per part."""
        parts = self.__dict__.get("_row_parts", False)
        if parts is False:
            parts = None
            big = [m for m in self.matrices
                   if isinstance(m, SparseMatrix) and m._dev().data.numel() >= _spm.PART_NNZ]
            if big:
                n = self.shape[0]
                lead = max(big, key=lambda m: m._dev().data.numel())
                ptr = lead._dev().indptr
                nnz = int(ptr[-1].item())
                k = -(-nnz // max(1, int(0.8 * _spm.PART_NNZ)))
                while True:
                    targets = torch.arange(1, k, device=ptr.device, dtype=torch.int64) * nnz // k
                    cuts = [0] + torch.searchsorted(ptr, targets).clamp_(0, n).tolist() + [n]
                    cuts = sorted(set(int(c) for c in cuts))
                    ok = all(int((m._dev().indptr[cuts[1:]] - m._dev().indptr[cuts[:-1]]).max().item())
                             < _spm.PART_NNZ for m in big)
                    if ok or k >= n:
                        break
                    k += 1
                parts = [(a, b, self[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            self.__dict__["_row_parts"] = parts
        return parts

    def _sandwich_parts(self, parts, d, rows, cols_host, colsum, center=None):
        """Sum of the parts' sandwiches (and of their column sums; with `center` the column sums of the centred
        blocks are handed on RAW: a part's centred sum + centre * the part's sum of weights)."""
        out = None
        for a, b, part in parts:
            r = None
            if rows is not None:
                r64 = rows.to(torch.int64)
                sel = r64[(r64 >= a) & (r64 < b)]
                if sel.numel() == 0:
                    continue
                r = (sel - a).to(torch.int32)
            cs = [None] * len(self.matrices) if colsum is not None else None
            sub_cen = None if center is None else _Centering(center.vec)
            res = part._sandwich_dev(d[a:b], r, cols_host, None, cs, center=sub_cen)
            if sub_cen is not None:
                center.groups = sub_cen.groups          # (the same path in every part: same selection)
            if sub_cen is not None and colsum is not None:
                dsum = (d[a:b] if r is None else d[a:b][r.to(torch.int64)]).sum(dtype=torch.float64)
                _, sub_p, _ = self._sandwich_plan(cols_host)
                for i in sub_cen.cs_centered:
                    # (a block the narrow path marked without a centre of its own -- sparse columns, dense
                    # blocks whose centres are all zero -- has centre 0: its centred sum IS the raw sum)
                    if cs[i] is not None and sub_cen.get(i) is not None:
                        cv = sub_cen.get(i).to(torch.float64)
                        if sub_p[i] is not None:
                            cv = cv[sub_p[i].to(torch.int64)]
                        cs[i] = cs[i].to(torch.float64) + cv * dsum
            out = res if out is None else out + res
            if colsum is not None:
                for i, c in enumerate(cs):
                    if c is None:
                        colsum[i] = False            # not available for every part: recompute
                    elif colsum[i] is not False:
                        colsum[i] = c if colsum[i] is None else colsum[i] + c
        if colsum is not None:
            for i, c in enumerate(colsum):
                if c is False:
                    colsum[i] = None
        if out is None:
            _, _, n_cols = self._sandwich_plan(cols_host)
            out = D.zeros((n_cols, n_cols), torch.float64)
        return out

    def _full_product_pays(self, sub_h) -> bool:
        """May a column selection be computed as the UNRESTRICTED product followed by a selection of rows and
        columns of the result?  Only when that wastes little: the blocks without any selected column (which
        the reference's restricted loops, and the restricted path here, skip altogether -- a glum active set
        that leaves out a high-cardinality categorical or a wide sparse block) must be a small share of the
        product's work, and the full (p, p) float64 result must be small (a categorical with 1e5 levels
        would make it 80 GB).  Work per block ~ the entries a row pass touches: dense columns, mean nonzeros
        per row, one code (+ its share of the level tables) per categorical."""
        p = self.shape[1]
        if p * p * 8 > FULL_RESULT_MAX_BYTES:
            return False
        n = max(self.shape[0], 1)
        tot = unsel = 0.0
        for b, mb in enumerate(self.matrices):
            if isinstance(mb, DenseMatrix):
                wb = float(mb.shape[1])
            elif isinstance(mb, SparseMatrix):
                nnz = mb._devblk.data.numel() if mb._devblk is not None else mb._host().nnz
                wb = max(1.0, float(nnz) / n)
            else:
                wb = 1.0 + mb.shape[1] / 256.0
            tot += wb
            if len(sub_h[b]) == 0:
                unsel += wb
        return unsel <= FULL_UNSELECTED_SHARE * tot

    def _full_product_pays_cols(self, cols_host) -> bool:
        key = np.asarray(cols_host).tobytes()
        hit = self.__dict__.get("_fpp_cache")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_fpp_cache"] = (key, self._full_product_pays(self._split_col_subsets(cols_host)[1]))
        return hit[1]

    def _narrow_plan(self, cols_host):
        """Bookkeeping of the dense-block form of a narrow column selection (see NARROW_COLS), or
        None when it does not apply.  The last selection is cached."""
        key = np.asarray(cols_host).tobytes()
        hit = self.__dict__.get("_narrow_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        nar = None
        pos_h, sub_h, n_cols = self._split_col_subsets(cols_host)
        mats = self.matrices
        noncat = [b for b, mb in enumerate(mats) if not isinstance(mb, CategoricalMatrix)]
        # (a categorical without a selected column takes no part: the reference's restricted loops skip it,
        # and its levels would only enlarge the intermediate result)
        cat = [b for b, mb in enumerate(mats) if isinstance(mb, CategoricalMatrix) and len(sub_h[b]) > 0]
        w = sum(len(sub_h[b]) for b in noncat)
        itemsize = np.dtype(self.dtype).itemsize
        if (noncat and 0 < w <= NARROW_COLS and self.shape[0] > 0
                and all(isinstance(mats[b], (DenseMatrix, SparseMatrix)) and mats[b].dtype == self.dtype
                        for b in noncat)
                and any(len(sub_h[b]) < mats[b].shape[1] for b in noncat)
                and self.shape[0] * (w + 1) * itemsize * 4 < torch.cuda.mem_get_info()[0]):
            w_pad = w + (w & 1)               # even width: 16-byte rows for the categorical x dense kernel
            t0, parts = 0, []
            sel = np.zeros(n_cols, dtype=np.int64)
            for b in noncat:
                s = len(sub_h[b])
                if s == 0:
                    continue
                sc = np.asarray(sub_h[b], dtype=np.int32)
                if isinstance(mats[b], SparseMatrix):
                    # from the CSC form (built once per block, a device sort): only the selected
                    # columns' entries are read -- 1.4 -> 0.1 ms at cfg4 / 5 % against a CSR pass
                    rws, vls, bstart, _, col_bptr = mats[b]._dev().csc_blocks()
                    cd = D.idx_dev(sc, torch.int64)
                    seg = torch.stack([bstart[col_bptr[cd]], bstart[col_bptr[cd + 1]]], dim=1).contiguous()
                    mx = int((seg[:, 1] - seg[:, 0]).max().item()) if s else 0
                    tcol = D.to_dev(t0 + np.arange(s, dtype=np.int32))
                    parts.append((b, t0, s, (rws, vls, seg, tcol, mx)))
                else:
                    parts.append((b, t0, s, D.to_dev(sc)))
                sel[np.asarray(pos_h[b], dtype=np.int64)] = t0 + np.arange(s)
                t0 += s
            off, cat_off, cat_sub = w_pad, {}, {}
            for i in cat:
                cat_off[i] = off
                if len(sub_h[i]):
                    sel[np.asarray(pos_h[i], dtype=np.int64)] = off + np.asarray(sub_h[i], dtype=np.int64)
                    cat_sub[i] = D.idx_dev(sub_h[i], torch.int64)
                off += mats[i].shape[1]
            indices = [np.arange(w_pad)] + [cat_off[i] + np.arange(mats[i].shape[1]) for i in cat]
            nar = dict(w=w, w_pad=w_pad, parts=parts, cat=cat, cat_sub=cat_sub, indices=indices,
                       sel=D.to_dev(sel), tmp=None, n_cols=n_cols)
        elif noncat and w > NARROW_COLS and FULL_THEN_SELECT <= 1.0 and self._full_product_pays(sub_h):
            nar = "wide"
        self.__dict__["_narrow_cache"] = (key, nar)
        return nar

    def _sandwich_narrow(self, nar, d, rows, colsum, center=None):
        """The sandwich under a narrow column selection: the selected dense / sparse columns as one
        dense block T, the unrestricted product of [T | categorical blocks], the selected rows and
        columns of that (small) result."""
        from .ext import dense as xd
        from .ext import sparse as xs
        from .ext._types import DenseDev

        n = self.shape[0]
        T = torch.zeros((n, nar["w_pad"]), dtype=d.dtype, device=d.device)
        for b, t0, s, idx in nar["parts"]:
            mb = self.matrices[b]
            if isinstance(mb, SparseMatrix):
                xs.csc_densify_cols(*idx, T)
            else:
                xd.dense_gather_cols(mb._dev(), idx, T, t0)
        tmp = nar["tmp"]
        if tmp is None:
            tmp = nar["tmp"] = SplitMatrix([DenseMatrix(T)] + [self.matrices[i] for i in nar["cat"]],
                                           nar["indices"])
        dm = tmp.matrices[0]
        dm._devblk = DenseDev(T, n, nar["w_pad"], 0)
        sub_cen = None
        if center is not None:
            # the centres of the selected dense columns, 0 for the written-out sparse columns and the padding
            tc = torch.zeros((nar["w_pad"],), dtype=d.dtype, device=d.device)
            for b, t0, s, idx in nar["parts"]:
                if center.get(b) is not None:
                    tc[t0:t0 + s] = center.get(b)[idx.to(torch.int64)]
            sub_cen = _Centering({0: tc})
        try:
            cs_tmp = [None] * len(tmp.matrices) if colsum is not None else None
            full = tmp._sandwich_dev(d, rows, None, None, cs_tmp, center=sub_cen)
        finally:
            dm._devblk = None                 # T is released with this call
        if colsum is not None:
            if cs_tmp[0] is not None:
                for b, t0, s, _ in nar["parts"]:
                    colsum[b] = cs_tmp[0][t0:t0 + s]
                    if sub_cen is not None and 0 in sub_cen.cs_centered:
                        center.cs_centered.add(b)
            # the categoricals' X'd falls out of the product whether or not anything is centred
            for k, i in enumerate(nar["cat"]):
                if cs_tmp[1 + k] is not None and i in nar["cat_sub"]:
                    colsum[i] = cs_tmp[1 + k][nar["cat_sub"][i]]
        if center is not None:
            # every entry between two columns of T came out of T's centred self term
            g = torch.full_like(nar["sel"], -1, dtype=torch.int32)
            g[nar["sel"] < nar["w_pad"]] = _Centering.NARROW_GROUP
            center.groups = g
        sel = nar["sel"]
        return full.index_select(0, sel).index_select(1, sel)

    def _sandwich_dev(self, d, rows, cols_host, plan=None, colsum=None, center=None):
        """d: device tensor; rows: int32 device tensor or None; cols_host: host list or None.
        Returns the float64 (n_cols, n_cols) device result (split_matrix.py:324-356).
        colsum: optional list (one slot per block) that receives X_block' d[rows] (restricted to
        the block's columns) wherever it falls out of the sandwich for free.
        center (_Centering or None): dense blocks whose SELF term is computed centred (the cross terms stay
        raw); it also records which column sums are centred."""
        if cols_host is not None and len(cols_host) >= FULL_THEN_SELECT * self.shape[1] \
                and len(cols_host) > 0 and self._full_product_pays_cols(cols_host):
            pos_d, sub_d, n_cols = plan if plan is not None else self._sandwich_plan(cols_host)
            cs_full = [None] * len(self.matrices) if colsum is not None else None
            full = self._sandwich_dev(d, rows, None, None, cs_full, center=center)
            if colsum is not None:
                for i, c in enumerate(cs_full):
                    if c is not None:
                        colsum[i] = c if sub_d[i] is None else c[sub_d[i].to(torch.int64)]
            cd = self._cols_dev64(cols_host)
            return full.index_select(0, cd).index_select(1, cd)
        parts = self._parts()
        if parts is not None:
            return self._sandwich_parts(parts, d, rows, cols_host, colsum, center)
        if cols_host is not None and plan is None and NARROW_COLS > 0:
            nar = self._narrow_plan(cols_host)
            if nar == "wide":
                # more selected dense + sparse columns than the dense-block form takes: the generic
                # restricted kernels stream everything and cost the full product or more (3.35-3.7
                # vs 3.5 ms at 2M rows, profiles/r3_cols_rows.txt) -- the tuned unrestricted
                # product + selection is never slower, so the cost is monotone in the selection
                cs_full = [None] * len(self.matrices) if colsum is not None else None
                full = self._sandwich_dev(d, rows, None, None, cs_full, center=center)
                if colsum is not None:
                    _, sub_sel, _ = self._sandwich_plan(cols_host)
                    for i, c in enumerate(cs_full):
                        if c is not None:
                            colsum[i] = c if sub_sel[i] is None else c[sub_sel[i].to(torch.int64)]
                cd = self._cols_dev64(cols_host)
                return full.index_select(0, cd).index_select(1, cd)
            if nar is not None and d.dtype == D.torch_dtype(self.dtype):
                return self._sandwich_narrow(nar, d, rows, colsum, center)
        pos_d, sub_d, n_cols = plan if plan is not None else self._sandwich_plan(cols_host)
        out = D.zeros((n_cols, n_cols), torch.float64)
        mats = self.matrices
        empty = [sd is not None and D.nlen(sd) == 0 for sd in sub_d]
        done = set()
        return self._sandwich_terms(d, rows, cols_host, colsum, out, pos_d, sub_d, empty, done, center)

    def _sandwich_terms(self, d, rows, cols_host, colsum, out, pos_d, sub_d, empty, done, center=None):
        """The block products of one sandwich, in line on the current stream."""
        from .ext import dense as xd

        mats = self.matrices
        # ---- fused categorical cross terms: one pass over the dense / sparse block serves
        #      every categorical block (tm_multi_cat_*), instead of one pass per pair
        cat_ids = [i for i, m in enumerate(mats) if isinstance(m, CategoricalMatrix) and not empty[i]
                   and m.shape[1] > 0]
        budget = (128 * 1024) // 8       # LDS tiles are made of doubles for float32 data too
        groups = self._cat_groups(cat_ids)
        if groups:
            d_eff = D.masked_d(d, rows)   # row restriction = masked d (excluded rows contribute 0)
        for grp in groups:
            cats = [(mats[i]._dev(), mats[i].shape[1], mats[i].drop_first) for i in grp]
            total = sum(c[1] for c in cats)
            offs = np.concatenate([[0], np.cumsum([c[1] for c in cats])])
            for w, mw in enumerate(mats):
                if empty[w] or mw.dtype != self.dtype or d.dtype != D.torch_dtype(self.dtype):
                    continue
                stacked = self._fused_cats(mw, cats, grp, d_eff, rows, total, budget, d)
                if stacked is None:
                    continue
                # no column restriction: the stacked result goes out in ONE scatter (24
                # categoricals x 2 operands were 48 launches of ~5 us)
                whole = cols_host is None and len(grp) > 1
                if whole:
                    cache = self.__dict__.setdefault("_group_pos", {})
                    gpos = cache.get(tuple(grp))
                    if gpos is None:
                        gpos = cache[tuple(grp)] = torch.cat([pos_d[i] for i in grp])
                    xsplit.scatter_block(stacked.contiguous(), gpos, pos_d[w], out, mirror=True)
                for ci, i in enumerate(grp):
                    res = stacked[int(offs[ci]):int(offs[ci + 1])]
                    if (colsum is not None and colsum[w] is None and not mats[i].drop_first
                            and not mats[i]._has_missings):
                        cs = res.sum(dim=0)           # all levels of a complete categorical
                        colsum[w] = cs if sub_d[w] is None else cs[sub_d[w].to(torch.int64)]
                    if not whole:
                        res = CategoricalMatrix._restrict(res, sub_d[i], sub_d[w])
                        xsplit.scatter_block(res.contiguous(), pos_d[i], pos_d[w], out, mirror=True)
                    done.add((min(i, w), max(i, w)))
        # ---- categoricals with too many levels for the fused groups, against a NARROW dense block: the generic
        #      LDS-tile kernel holds [all their levels][8 columns or fewer] at once, so one pass over the block
        #      serves them all (the reference's design dense_cat: two 1000-level categoricals x 5 dense columns were
        #      two passes of 0.08 ms)
        big = [i for i in cat_ids if mats[i].shape[1] > self.FUSED_LEVELS]
        if len(big) >= 2 and len(big) <= 16 and d.dtype == D.torch_dtype(self.dtype):
            cats = [(mats[i]._dev(), mats[i].shape[1], mats[i].drop_first) for i in big]
            offs = np.concatenate([[0], np.cumsum([c[1] for c in cats])])
            d_big = None
            for w, mw in enumerate(mats):
                if (empty[w] or not isinstance(mw, DenseMatrix) or mw.dtype != self.dtype
                        or not xsplit.multi_cat_dense_tile_ok(cats, mw._dev_c())):
                    continue
                if d_big is None:
                    d_big = D.masked_d(d, rows)
                stacked = xsplit.multi_cat_dense_sandwich(cats, d_big, mw._dev_c())
                for ci, i in enumerate(big):
                    res = stacked[int(offs[ci]):int(offs[ci + 1])]
                    if (colsum is not None and colsum[w] is None and not mats[i].drop_first
                            and not mats[i]._has_missings):
                        cs = res.sum(dim=0)
                        colsum[w] = cs if sub_d[w] is None else cs[sub_d[w].to(torch.int64)]
                    res = CategoricalMatrix._restrict(res, sub_d[i], sub_d[w])
                    xsplit.scatter_block(res.contiguous(), pos_d[i], pos_d[w], out, mirror=True)
                    done.add((min(i, w), max(i, w)))
        # ---- all categorical x categorical tables that fit an LDS tile, and the categorical
        #      diagonals, in ONE pass over the codes (tm_multi_cat_pairs_*): a design with k
        #      categoricals has k (k - 1) / 2 of them, one launch each was ~30 us apiece
        cat_diag = {}
        diag_scattered = set()
        # TABMAT_AMD_DETERMINISTIC: the diagonals must come from the fixed-order histogram
        # (csrc/cat_det.hip), not from LDS-atomic tables -- no fused plan, no row-sum shortcut
        det = _cm.DETERMINISTIC
        plan = self._cat_pairs_plan() if CAT_PAIRS_FUSED and not det else None
        if (plan is not None and plan.n_pairs > 0
                and d.dtype in (torch.float32, torch.float64)
                and d.dtype == D.torch_dtype(self.dtype)):
            cl = [(mats[i]._dev(), mats[i].shape[1], mats[i].drop_first) for i in plan.cat_ids]
            # column restriction: the full tables are accumulated, the scatter drops the
            # unselected levels (position -1)
            pos_sel = None if cols_host is None else self._pairs_pos_sel(plan, cols_host)
            tables = xsplit.multi_cat_pairs(plan, cl, d, rows, out, pos=pos_sel)
            for i, j, toff, li, lj, stride in plan.pairs:
                if i == j:
                    if not empty[i]:
                        full = tables[toff:toff + (li - 1) * stride + 1:stride]
                        if sub_d[i] is not None and colsum is not None:
                            full = full[sub_d[i].to(torch.int64)]
                        cat_diag[i] = full.to(d.dtype)
                        diag_scattered.add(i)
                else:
                    done.add((i, j))
        # the other categorical x categorical tables: the diagonal of a categorical block is the
        # row sum of its table with any COMPLETE partner (every row has exactly one level there),
        # which saves its histogram pass
        complete = [isinstance(m, CategoricalMatrix) and not m.drop_first and not m._has_missings
                    and sub_d[k] is None and not empty[k] for k, m in enumerate(mats)]
        for i, mi in enumerate(mats):
            if empty[i] or not isinstance(mi, CategoricalMatrix):
                continue
            for j in range(i + 1, len(mats)):
                if empty[j] or not isinstance(mats[j], CategoricalMatrix) or (i, j) in done:
                    continue
                res = mi._cross_sandwich_dev(mats[j], d, rows, sub_d[i], sub_d[j])
                xsplit.scatter_block(res.contiguous(), pos_d[i], pos_d[j], out, mirror=True)
                done.add((i, j))
                if DIAG_FROM_PAIRS and not det and complete[j] and i not in cat_diag:
                    cat_diag[i] = res.sum(dim=1)
                if DIAG_FROM_PAIRS and not det and complete[i] and j not in cat_diag:
                    cat_diag[j] = res.sum(dim=0)
        # self terms first, then the remaining cross terms
        for i, mi in enumerate(mats):
            if empty[i]:
                continue
            if isinstance(mi, CategoricalMatrix):
                diag = cat_diag[i] if i in cat_diag else mi._sandwich_diag_dev(d, rows, sub_d[i])
                if colsum is not None:
                    colsum[i] = diag          # one-hot entries are 0 / 1: C' d = diag(C' D C)
                if i not in diag_scattered:
                    xsplit.scatter_block(diag, pos_d[i], pos_d[i], out, diag=True)
            elif (colsum is not None and colsum[i] is None and isinstance(mi, DenseMatrix)
                  and rows is None and sub_d[i] is None
                  and (both := mi._sandwich_xtd_dev(d, cen_i := (center.get(i) if center else None))) is not None):
                # X_dense' d comes out of the syrk's own pass (csrc/syrk_i8.hip, csrc/syrk_co.hip)
                res, colsum[i] = both
                if cen_i is not None:
                    center.cs_centered.add(i)
                xsplit.scatter_block(res, pos_d[i], pos_d[i], out)
            elif isinstance(mi, DenseMatrix) and center is not None and center.get(i) is not None:
                res = mi._sandwich_dev(d, rows, sub_d[i], center=center.get(i))
                xsplit.scatter_block(res, pos_d[i], pos_d[i], out)
            else:
                res = mi._sandwich_dev(d, rows, sub_d[i])
                xsplit.scatter_block(res, pos_d[i], pos_d[i], out)
        for i, mi in enumerate(mats):
            if empty[i]:
                continue
            for j in range(i + 1, len(mats)):
                if empty[j] or (i, j) in done:
                    continue
                # sparse x dense: X_sparse' d rides along in the gather's stream loop
                si, di = (i, j) if isinstance(mi, SparseMatrix) else (j, i)
                if (colsum is not None and colsum[si] is None and isinstance(mats[si], SparseMatrix)
                        and isinstance(mats[di], DenseMatrix)):
                    box = []
                    res = mats[si]._cross_sandwich_dev(mats[di], d, rows, sub_d[si], sub_d[di],
                                                       colsum_box=box)
                    if box:
                        colsum[si] = box[0]
                    if si != i:
                        res = res.T
                else:
                    res = mi._cross_sandwich_dev(mats[j], d, rows, sub_d[i], sub_d[j])
                xsplit.scatter_block(res.contiguous(), pos_d[i], pos_d[j], out, mirror=True)
        return out

    def sandwich_graph(self, d, rows=None, cols=None):
        """HIP-graph form of `sandwich` for solvers that call it with the same `rows` / `cols`
        every iteration: returns f(d) -> float64 (n_cols, n_cols) device tensor (a static buffer,
        overwritten by the next call) that replays the captured launch sequence of
        `_sandwich_dev` (tabmat_amd/graph.py).  d: a device tensor of the matrix dtype."""
        from .graph import CapturedProduct

        check_sandwich_compatible(self, d)
        rows_d = D.idx_dev(normalize_index(rows, self.shape[0]))
        cols_n = collapse_identity(normalize_index(cols, self.shape[1]), self.shape[1])
        plan = self._sandwich_plan(cols_n)      # index uploads happen here, outside the capture
        return CapturedProduct(lambda dd: self._sandwich_dev(dd, rows_d, cols_n, plan), d)

    def sandwich(self, d, rows=None, cols=None):
        """X[rows, cols].T @ diag(d[rows]) @ X[rows, cols]; always float64 like the reference
        (split_matrix.py:336)."""
        on_dev = D.is_dev(d)
        if not on_dev:
            d = np.asarray(d)
        check_sandwich_compatible(self, d)
        rows_n = normalize_index(rows, self.shape[0])
        cols_n = collapse_identity(normalize_index(cols, self.shape[1]), self.shape[1])
        out = self._sandwich_dev(D.to_dev(d), D.idx_dev(rows_n), cols_n)
        return out if on_dev else D.to_host(out)

    def matvec(self, v, cols=None, out=None):
        """split_matrix.py:373-420."""
        assert not sps.issparse(v)
        on_dev = D.is_dev(v)
        if not on_dev:
            v = np.asarray(v)
        check_matvec_dimensions(self, v, transpose=False)
        check_matvec_out_shape(self, out)
        if v.ndim > 1 and any(isinstance(m, CategoricalMatrix) for m in self.matrices):
            raise NotImplementedError(
                "CategoricalMatrix.matvec is only implemented for 1d arrays.")
        cols_n = collapse_identity(normalize_index(cols, self.shape[1]), self.shape[1])
        tdt = D.torch_dtype(self.dtype)
        v_dev = D.to_dev(v, tdt)
        if cols_n is not None and len(cols_n) >= FULL_THEN_SELECT_MV * self.shape[1] and len(cols_n) > 0 \
                and self._blocks_finite():      # (0 x inf of an excluded column would leak a NaN)
            cd = self._cols_dev64(cols_n)           # X[:, cols] v[cols] = X (v with zeros elsewhere)
            vm = torch.zeros_like(v_dev)
            vm[cd] = v_dev[cd]
            v_dev, cols_n = vm, None
        _, sub_d, _ = self._sandwich_plan(cols_n)
        idx_d = self._full_dev_indices()
        if v_dev.ndim == 1:
            res = D.zeros((self.shape[0],), tdt)
            fused = set()
            plan = self._cat_hist_plan() if CAT_PAIRS_FUSED else None
            if plan is not None:
                # all categorical blocks in ONE pass (one gather + one launch per block before:
                # 20 categoricals 0.33 ms for 0.18 GB); a column selection becomes zeros in the
                # coefficient vector these blocks read
                cl = [(self.matrices[i]._dev(), self.matrices[i].shape[1], self.matrices[i].drop_first)
                      for i in plan.cat_ids]
                vm = v_dev
                if cols_n is not None:
                    cd = self._cols_dev64(cols_n)
                    vm = torch.zeros_like(v_dev)
                    vm[cd] = v_dev[cd]
                xsplit.multi_cat_matvec(plan, cl, vm, res)
                fused = set(plan.cat_ids)
            for bi, (mat, idx, scd) in enumerate(zip(self.matrices, idx_d, sub_d)):
                if (scd is not None and D.nlen(scd) == 0) or bi in fused:
                    continue
                vb = v_dev[idx]
                if isinstance(mat, CategoricalMatrix):
                    mat._matvec_dev(vb, scd, res)
                else:
                    if scd is not None and D.nlen(scd) == mat.shape[1]:
                        scd = None
                    mat._matvec_dev(vb, None, scd, res, False)
        else:
            # 2-D operand (no categorical block here): one multi-right-hand-side launch per block
            from .ext import dense as xd
            from .ext import sparse as xs

            res = D.zeros((self.shape[0], v_dev.shape[1]), tdt)
            for mat, idx, scd in zip(self.matrices, idx_d, sub_d):
                if scd is not None and D.nlen(scd) == 0:
                    continue
                if scd is not None and D.nlen(scd) == mat.shape[1]:
                    scd = None
                vb = v_dev[idx]
                if isinstance(mat, DenseMatrix):
                    res += xd.dense_matvec_multi(mat._dev(), vb, None, scd, False)
                else:
                    res += xs.csr_matvec_multi(mat._dev(), vb, None, scd, False)
        if not on_dev:
            res = D.to_host(res)
            if np.issubdtype(v.dtype, np.floating):
                res = res.astype(np.result_type(self.dtype, v.dtype), copy=False)
        if out is None:
            return res
        out += res
        return out

    def transpose_matvec(self, v, rows=None, cols=None, out=None):
        """split_matrix.py:422-460."""
        on_dev = D.is_dev(v)
        if not on_dev:
            v = np.asarray(v)
        check_matvec_dimensions(self, v, transpose=True)
        check_transpose_matvec_out_shape(self, out)
        if v.ndim > 1 and any(isinstance(m, CategoricalMatrix) for m in self.matrices):
            raise NotImplementedError(
                "CategoricalMatrix.transpose_matvec is only implemented for 1d arrays.")
        rows_n = normalize_index(rows, self.shape[0])
        cols_n = collapse_identity(normalize_index(cols, self.shape[1]), self.shape[1])
        select = None
        if cols_n is not None and len(cols_n) >= FULL_THEN_SELECT_MV * self.shape[1] and len(cols_n) > 0:
            select, cols_n = cols_n, None           # all columns, the selection picked at the end
        pos_d, sub_d, n_cols = self._sandwich_plan(cols_n)     # device index arrays (last selection cached)
        tdt = D.torch_dtype(self.dtype)
        v_dev = D.to_dev(v, tdt)
        rd = D.idx_dev(rows_n)
        if rows_n is not None and len(rows_n) == self.shape[0]:
            rd = None
        if v_dev.ndim == 1:
            res = D.zeros((n_cols,), tdt)
            empty_rows = rows_n is not None and len(rows_n) == 0
            fused = set()
            plan = self._cat_hist_plan() if (CAT_PAIRS_FUSED and not empty_rows and n_cols > 0
                                             and not _cm.DETERMINISTIC) else None
            if plan is not None and plan.n_pairs > 0:
                # every categorical block's histogram from ONE pass over the codes (one launch per
                # block was ~35 us each: 20 categoricals 0.69 ms for 0.18 GB); a column selection
                # picks its entries out of the full-length result
                cl = [(self.matrices[i]._dev(), self.matrices[i].shape[1], self.matrices[i].drop_first)
                      for i in plan.cat_ids]
                direct = cols_n is None and tdt == torch.float64
                tgt = res if direct else D.zeros((self.shape[1],), torch.float64)
                xsplit.multi_cat_pairs(plan, cl, v_dev, rd, tgt, vector=True)
                if not direct:
                    if cols_n is not None:
                        tgt = tgt[self._cols_dev64(cols_n)]
                    res += tgt.to(tdt)
                fused = {i for i, *_ in plan.pairs}
            for bi, (mat, pd, scd) in enumerate(zip(self.matrices, pos_d, sub_d)):
                if empty_rows or (scd is not None and D.nlen(scd) == 0) or bi in fused:
                    continue
                if isinstance(mat, CategoricalMatrix):
                    full = D.zeros((mat.shape[1],), tdt)
                    mat._transpose_matvec_dev(v_dev, rd, scd, full)
                    part = full if scd is None else full[scd.to(torch.int64)]
                else:
                    if scd is not None and D.nlen(scd) == mat.shape[1]:
                        scd = None  # sorted unique columns covering the block = all of them
                    part = mat._matvec_dev(v_dev, rd, scd, None, True)
                res[pd] += part
        else:
            from .ext import dense as xd
            from .ext import sparse as xs

            res = D.zeros((n_cols, v_dev.shape[1]), tdt)
            empty_rows = rows_n is not None and len(rows_n) == 0
            for mat, pd, scd in zip(self.matrices, pos_d, sub_d):
                if empty_rows or (scd is not None and D.nlen(scd) == 0):
                    continue
                if scd is not None and D.nlen(scd) == mat.shape[1]:
                    scd = None
                if isinstance(mat, DenseMatrix):
                    res[pd] += xd.dense_matvec_multi(mat._dev(), v_dev, rd, scd, True)
                else:
                    res[pd] += xs.csr_matvec_multi(mat._dev(), v_dev, rd, scd, True)
        if select is not None:
            res, cols_n = res[self._cols_dev64(select)], select
        if not on_dev:
            res = D.to_host(res)
            if np.issubdtype(v.dtype, np.floating):
                res = res.astype(np.result_type(self.dtype, v.dtype), copy=False)
        if out is None:
            return res
        if cols_n is None:
            out += res
        else:
            out[cols_n if not D.is_dev(out) else D.idx_dev(cols_n, torch.int64)] += res
        return out
