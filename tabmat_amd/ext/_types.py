"""Device-resident block storage handed to the ext functions."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

from .. import _device as D


@dataclass
class DenseDev:
    """Dense block in HBM.  order_f=0: buf is the C-ordered (n, m) array; order_f=1: buf holds
    the F-ordered array, i.e. the C-ordered (m, n) transpose (dense_matrix.py:47-58 keeps
    whichever contiguity the caller supplied)."""

    buf: torch.Tensor
    n: int
    m: int
    order_f: int

    @property
    def dtype(self):
        return self.buf.dtype

    @staticmethod
    def from_host(X: np.ndarray) -> "DenseDev":
        n, m = X.shape
        if X.flags["C_CONTIGUOUS"]:
            return DenseDev(D.to_dev(X), n, m, 0)
        if X.flags["F_CONTIGUOUS"]:
            return DenseDev(D.to_dev(X.T), n, m, 1)
        raise Exception("The matrix X is not contiguous.")  # ext/dense.pyx:43

    @staticmethod
    def from_tensor(t: torch.Tensor) -> "DenseDev":
        """(n, m) cuda tensor; C-contiguous or the .T of a contiguous (m, n) tensor."""
        n, m = t.shape
        if t.is_contiguous():
            return DenseDev(t, n, m, 0)
        if t.T.is_contiguous():
            return DenseDev(t.T, n, m, 1)
        return DenseDev(t.contiguous(), n, m, 0)

    def as_2d(self) -> torch.Tensor:
        return self.buf if self.order_f == 0 else self.buf.T


# Column indices of the CSR twin: int32 as uploaded, or -- once to_device() has built the twins (compact_indices) and
# the block has at most 65 536 columns -- ONLY the 16-bit copy the unrestricted matvec / transpose_matvec kernels
# stream (1.0 GB less at BASELINE configs[3]).  The generic / restricted entry points and the twin
# builders read `.indices`, which then widens into a library-wide scratch tensor (the two most recent widenings are
# kept; released by later ones or by release_index_scratch()).
COMPACT_CSR_INDICES = os.environ.get("TABMAT_AMD_COMPACT_CSR", "1") != "0"
# K2b's static block list with 12-byte descriptors (blocks of fewer than 2^24 rows; tm_sparse_sandwich_blocks_p12_*)
K2B_DESC12 = os.environ.get("TABMAT_AMD_K2B_DESC12", "1") != "0"
_WIDE = []          # [(weak reference to the owner CsrDev, int32 tensor)], most recent last; at most two (a call may name two blocks)


def release_index_scratch() -> None:
    del _WIDE[:]


class CsrDev:
    """CSR twin of a sparse block in HBM (sparse_matrix.py:133-143): int32 column indices,
    int64 indptr (ext/sparse.pyx:13-15 accepts int32 or int64; narrowed/widened on upload)."""

    def __init__(self, data, indices, indptr, n, m):
        self.data = data
        self._ind32 = indices
        self._ind16 = None
        self.indptr = indptr
        self.n = int(n)
        self.m = int(m)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def indices(self) -> torch.Tensor:
        """int32 column indices (widened from the 16-bit twin into the shared scratch once compacted)."""
        if self._ind32 is not None:
            return self._ind32
        import weakref

        _WIDE[:] = [(o, t) for o, t in _WIDE if o() is not None]      # (scratch of blocks that are gone)
        for k, (owner, t) in enumerate(_WIDE):
            if owner() is self:
                _WIDE.append(_WIDE.pop(k))
                return t
        while len(_WIDE) >= 2:
            _WIDE.pop(0)                                       # (free the oldest scratch before allocating)
        t = self._ind16.to(torch.int32).bitwise_and_(0xFFFF)
        _WIDE.append((weakref.ref(self), t))
        return t

    def indices16(self):
        """uint16 twin of the column indices (bit pattern in an int16 tensor) for the unrestricted matvec /
        transpose_matvec stream kernels (tm_csr_{matvec,rmatvec}_u16_*): 10 instead of 12 bytes per entry."""
        i16 = self._ind16
        if i16 is None:
            assert self.m <= 65536
            w = self._ind32
            i16 = self._ind16 = torch.where(w >= 32768, w - 65536, w).to(torch.int16).contiguous()
        return i16

    def compact_indices(self) -> bool:
        """Keep only the 16-bit column indices (blocks of at most 65 536 columns).  Called by
        SparseMatrix.to_device() after the twins are built."""
        if (COMPACT_CSR_INDICES and self._ind32 is not None and self.m <= 65536
                and self._ind32.numel() > 0):
            self.indices16()
            self._ind32 = None
            return True
        return False

    def nbytes(self) -> int:
        """HBM bytes of the CSR arrays and of every twin built so far."""
        seen, tot = set(), 0

        def add(t):
            nonlocal tot
            if isinstance(t, torch.Tensor):
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    tot += t.numel() * t.element_size()
            elif isinstance(t, (tuple, list)):
                for x in t:
                    add(x)

        for v in self.__dict__.values():
            add(v)
        return tot

    def chunk_col8(self):
        """uint8 [nnz]: the column of every chunk-major entry inside its column chunk (tm_sparse_sandwich_blocks_u8_*,
        the row-list kernels)."""
        return self.chunk_major()[1]

    def chunk_cols32(self):
        """int32 [nnz] block columns of the chunk-major entries -- NOT kept since round 6 (the byte columns serve every
        kernel): rebuilt on request for the int32 entry points (K2B_U8 = False, the record twins' builders)."""
        _, c8, cptr = self.chunk_major()
        from .._lib import lib

        ch = int(lib().tm_sparse_chunk_cols())
        nch = int(cptr.shape[0])
        sizes = (cptr[:, -1] - cptr[:, 0]).to(torch.int64)
        base = torch.repeat_interleave(torch.arange(nch, device=c8.device, dtype=torch.int32) * ch, sizes,
                                       output_size=int(c8.numel()))
        return base.add_(c8.to(torch.int32))

    def chunk_major(self):
        """(cm_data, cm_col8 uint8, cptr int32 [NCH, n + 1]): the entries regrouped by column chunk,
        inside a chunk by row (see tm_sparse_sandwich_chunked_*); cm_col8 = the column INSIDE the chunk.
        Built once per block (ingest: one device key sort), cached."""
        cm = getattr(self, "_cm", None)
        if cm is None:
            from .._lib import lib

            ch = int(lib().tm_sparse_chunk_cols())
            assert ch <= 256
            nch = max(1, (self.m + ch - 1) // ch)
            nnz = int(self.data.numel())
            if nnz >= 2**31:
                raise ValueError("chunk pointers need nnz < 2^31")
            dev = self.data.device
            ind = self.indices
            counts = self.indptr[1:] - self.indptr[:-1]
            rows = torch.repeat_interleave(torch.arange(self.n, device=dev, dtype=torch.int64), counts)
            chunk = torch.div(ind, ch, rounding_mode="floor")
            # CSR is row-major with ascending columns: a STABLE sort by the chunk number alone is
            # the (chunk, row, column) order -- an 8- or 16-bit radix sort instead of a 64-bit one
            small = chunk.to(torch.uint8 if nch <= 255 else torch.int16 if nch < 2**15 else torch.int32)
            perm = torch.sort(small, stable=True).indices
            del small
            cm_data = self.data[perm].contiguous()
            cm_col8 = torch.remainder(ind, ch).to(torch.uint8)[perm].contiguous()
            del perm, ind
            key = chunk.to(torch.int64) * self.n + rows
            del rows, chunk
            per = torch.bincount(key, minlength=nch * self.n) if nnz else \
                torch.zeros(nch * self.n, dtype=torch.int64, device=dev)
            del key
            start = (torch.cumsum(per, dim=0) - per).view(nch, self.n)
            ends = torch.cat([start[1:, 0], torch.tensor([nnz], device=dev, dtype=torch.int64)])
            cptr = torch.cat([start, ends[:, None]], dim=1).to(torch.int32).contiguous()
            cm = (cm_data, cm_col8, cptr)
            self._cm = cm
        return cm

    def chunk_records(self):
        """cm_rec int32 [nnz, 4]: one 16-byte record {value, column, row} per entry of the chunk-major twin
        (f64: value as two words; f32: {value bits, column, row, 0}) for tm_sparse_sandwich_pairs_*.  Built on first
        use (the rows from the chunk pointers: inside a chunk the entries are in row order), cached.  (Filled
        column by column: a 2-D gather of more than 2^32 bytes came back scrambled on this stack.)"""
        cr = getattr(self, "_cm_rec", None)
        if cr is None:
            cm_data, _, cptr = self.chunk_major()
            cm_ind = self.chunk_cols32()
            dev = cptr.device
            nnz = int(cm_data.numel())
            ar = torch.arange(self.n, device=dev, dtype=torch.int32)
            # (one record of tail padding, as the packed form has: the kernel's clamped look-ahead may name entry nnz)
            cr = torch.zeros((nnz + 1, 4), dtype=torch.int32, device=dev)[:nnz]
            f64 = cm_data.dtype == torch.float64
            if f64:
                words = cm_data.view(torch.int32).view(nnz, 2)
                cr[:, 0] = words[:, 0]
                cr[:, 1] = words[:, 1]
                cr[:, 2] = cm_ind
            else:
                cr[:, 0] = cm_data.view(torch.int32)
                cr[:, 1] = cm_ind
            pos = 0
            for c in range(int(cptr.shape[0])):
                k = int(cptr[c, -1].item()) - int(cptr[c, 0].item())
                if k:
                    cnt = (cptr[c, 1:] - cptr[c, :-1]).to(torch.int64)
                    cr[pos:pos + k, 3 if f64 else 2] = torch.repeat_interleave(ar, cnt, output_size=k)
                pos += k
            self._cm_rec = cr
        return cr

    def chunk_records_packed(self):
        """int32 [nnz * W + 4] (W = 3 for f64, 2 for f32): packed records {value, row << 7 | column inside the chunk}
        for tm_sparse_sandwich_pairs_pk_* (n < 2^25), or None."""
        cr = getattr(self, "_cm_rec_pk", False)
        if cr is False:
            cr = None
            if self.n < 2**25:
                from .._lib import lib

                ch = int(lib().tm_sparse_chunk_cols())
                cm_data, cm_c8, cptr = self.chunk_major()
                dev = cptr.device
                nnz = int(cm_data.numel())
                f64 = cm_data.dtype == torch.float64
                W = 3 if f64 else 2
                buf = torch.zeros(nnz * W + 4, dtype=torch.int32, device=dev)
                rec = buf[:nnz * W].view(nnz, W)
                if f64:
                    words = cm_data.view(torch.int32).view(nnz, 2)
                    rec[:, 0] = words[:, 0]
                    rec[:, 1] = words[:, 1]
                else:
                    rec[:, 0] = cm_data.view(torch.int32)
                ar = torch.arange(self.n, device=dev, dtype=torch.int64)
                pos = 0
                for c in range(int(cptr.shape[0])):
                    k = int(cptr[c, -1].item()) - int(cptr[c, 0].item())
                    if k:
                        cnt = (cptr[c, 1:] - cptr[c, :-1]).to(torch.int64)
                        rows = torch.repeat_interleave(ar, cnt, output_size=k)
                        word = (rows << 7) | cm_c8[pos:pos + k].to(torch.int64)
                        rec[pos:pos + k, W - 1] = torch.where(word >= 2**31, word - 2**32, word).to(torch.int32)
                    pos += k
                cr = buf
            self._cm_rec_pk = cr
        return cr

    def pair_blocks(self, n_wg: int = 0, nw: int = 16, cyclic=None, d12=None):
        """(blocks int32 [B, 4] -- or [B, 3]: 12-byte descriptors, see K2B_DESC12 --, wg_tab int32 [W, 8], max_nb): the static block list of
        tm_sparse_sandwich_blocks_* -- every (row, tile) of the chunk-major twin cut into blocks of
        at most 8 x 8 entries {first A entry, first B entry, row, nA | nB << 8 | flags}, tile after
        tile (part = I (I + 1) / 2 + J); the blocks of a tile are dealt to its workgroups by row range (`cyclic`
        rows per range, round-robin; 0 = one contiguous piece each; None / n_wg 0 = chosen by the list's length; workgroups in proportion to the block counts,
        ~n_wg in all); inside a workgroup the FULL
        blocks (both sides > 4 entries: 8 DPP steps) come first, then the HALF ones (4 steps; flag
        bits 16 / 17: the A / B side is the short one next to a long side, see csrc/sparse_blocks.hip).
        wg_tab row: {part, slot, first block, end, end of the FULL blocks, waves on the FULL list,
        first row, last row}.  Depends on the sparsity pattern only: built once, cached."""
        d12 = bool((K2B_DESC12 if d12 is None else d12) and self.n < 2**24)   # 12-byte descriptors: 24 bits of row
        cache = self.__dict__.get("_pb")
        if not isinstance(cache, dict):                      # (tests reset the cache with `_pb = None`)
            cache = self.__dict__["_pb"] = {}
        pb = cache.get(d12)
        if pb is None:
            _, _, cptr = self.chunk_major()
            nch, n = int(cptr.shape[0]), self.n
            dev = cptr.device
            cnt = (cptr[:, 1:] - cptr[:, :-1]).to(torch.int64)            # entries per (chunk, row)
            kb = torch.div(cnt + 7, 8, rounding_mode="floor")             # 8-entry pieces of a list
            rows_all = torch.arange(n, device=dev, dtype=torch.int64)
            per_row, counts = [], []
            for I in range(nch):
                for J in range(I + 1):
                    nb_row = kb[I] * (kb[I] + 1) // 2 if I == J else kb[I] * kb[J]
                    per_row.append(nb_row)
                    counts.append(int(nb_row.sum().item()))
            total_blocks = sum(counts)
            if total_blocks >= 2**31:
                raise ValueError("block list needs fewer than 2^31 blocks")
            # every workgroup's entry range must stay below 2^31 bytes (32-bit byte offsets relative to
            # the range's first entry)
            nnz = int(self.data.numel())
            min_nb = 1 + (nnz * 8) // (nch * 2**30)
            # waves per list in proportion to the block counts: a HALF step issues fewer instructions
            # than a FULL one (55 vs 88) but takes as long -- the kernel waits for its loads -- measured
            # at 4M rows: 1.64 ms with equal weights, 1.89 ms with 49 : 88 (profiles/r3_k2_blocks.txt)
            COST_FULL, COST_HALF, NW = 1.0, 1.0, int(nw)
            # four rounds of 256 workgroups and the round-robin deal pay for long lists only (10M rows: 4.09 ->
            # 3.96 ms; 1M rows: 0.39 -> 0.42 ms, profiles/r4_k2b.txt): below 50M blocks two rounds, contiguous pieces
            big = total_blocks >= 50_000_000
            if not n_wg:
                n_wg = 1024 if big else 512
            if cyclic is None:
                cyclic = 4096 if big else 0
            if cyclic and int((cptr[:, -1] - cptr[:, 0]).max().item()) * 8 >= 2**31 - 2**20:
                cyclic = 0          # a workgroup's range (a whole chunk) would no longer fit 32-bit byte offsets
            # workgroups per tile in proportion to its blocks, EXACTLY n_wg in all (largest remainders): the
            # grid then runs in whole rounds of 256 -- 386 workgroups (one and a half rounds) took 5.0 ms
            # where 256 take 4.25, 512 4.16 and 1024 4.04; with the round-robin deal of 4096-row ranges 3.99
            # (profiles/r4_k2b.txt)
            share = [n_wg * c / max(total_blocks, 1) for c in counts]
            cap = [max(1, -(-c // 256)) if c else 0 for c in counts]         # at least 256 blocks per workgroup
            nbp = [0 if c == 0 else max(min_nb, min(int(sh), cp)) for c, sh, cp in zip(counts, share, cap)]
            rest = n_wg - sum(nbp)
            for k in sorted(range(len(counts)), key=lambda k: share[k] - int(share[k]), reverse=True):
                if rest <= 0:
                    break
                if counts[k] and nbp[k] < cap[k]:
                    nbp[k] += 1
                    rest -= 1
            descs, tab, off, max_nb, part = [], [], 0, 1, -1
            for I in range(nch):
                for J in range(I + 1):
                    part += 1
                    c = counts[part]
                    if c == 0:
                        continue
                    nb_row = per_row[part]
                    row = torch.repeat_interleave(rows_all, nb_row)
                    start = torch.cumsum(nb_row, 0) - nb_row
                    idx = torch.arange(c, device=dev, dtype=torch.int64) - start[row]
                    if I == J:                                            # (a, b), b <= a: idx = a (a + 1) / 2 + b
                        a = torch.floor((torch.sqrt(8.0 * idx.to(torch.float64) + 1.0) - 1.0) * 0.5).to(torch.int64)
                        a = torch.where((a + 1) * (a + 2) // 2 <= idx, a + 1, a)
                        a = torch.where(a * (a + 1) // 2 > idx, a - 1, a)
                        b = idx - a * (a + 1) // 2
                    else:
                        kj = kb[J][row]
                        a = torch.div(idx, kj, rounding_mode="floor")
                        b = idx - a * kj
                    na = torch.clamp(cnt[I][row] - 8 * a, max=8)
                    nb = torch.clamp(cnt[J][row] - 8 * b, max=8)
                    full = (na > 4) & (nb > 4)
                    flags = torch.where((na <= 4) & (nb > 4), 1 << 16, 0) | torch.where((na > 4) & (nb <= 4), 1 << 17, 0)
                    if d12:
                        # {first A entry, first B entry, row | nA - 1 << 24 | nB - 1 << 27 | flags << 30}: 12 bytes
                        word = row | ((na - 1) << 24) | ((nb - 1) << 27) | ((flags >> 16) << 30)
                        word = torch.where(word >= 2**31, word - 2**32, word)
                        desc = torch.stack([cptr[I][row].to(torch.int64) + 8 * a, cptr[J][row].to(torch.int64) + 8 * b,
                                            word], dim=1).to(torch.int32)
                        del word
                    else:
                        desc = torch.stack([cptr[I][row].to(torch.int64) + 8 * a, cptr[J][row].to(torch.int64) + 8 * b,
                                            row, na | (nb << 8) | flags], dim=1).to(torch.int32)
                    nb_p = max(1, nbp[part])
                    if cyclic:
                        # row ranges of `cyclic` rows dealt round-robin: every workgroup of every tile sweeps the
                        # rows 0 -> n over the run of the kernel, so the ~5 tiles that read a chunk's entries
                        # of the same rows do so within a few ms of each other (Infinity Cache hits)
                        # (short matrices: at least four ranges per workgroup, or most of them would stay empty)
                        rng_rows = max(32, min(cyclic, n // (4 * nb_p)))
                        wg = torch.div(row, rng_rows, rounding_mode="floor") % nb_p
                    else:
                        per = -(-c // nb_p)
                        wg = torch.div(torch.arange(c, device=dev, dtype=torch.int64), per, rounding_mode="floor")
                    key = wg * 2 + (~full).to(torch.int64)
                    order = torch.sort(key, stable=True).indices          # row order kept inside a class
                    cnts = torch.bincount(key, minlength=nb_p * 2).view(nb_p, 2)
                    desc = desc[order]
                    # first / last row of every workgroup: the stable sort keeps the (ascending) row order inside a
                    # (workgroup, class) segment, so they are the rows at the segments' ends -- 2 nb_p reads.  (Round
                    # 5 asked scatter_reduce_ for them: millions of rows contending for nb_p addresses, 200 ms per
                    # tile, 2.2 of the 2.4 s this twin took to build at BASELINE configs[3].)
                    seg_end = torch.cumsum(cnts.reshape(-1), 0)
                    seg_lo = (seg_end - cnts.reshape(-1)).clamp_(max=c - 1)
                    seg_hi = (seg_end - 1).clamp_(min=0)
                    srow = row[order]
                    big = torch.iinfo(torch.int64).max
                    some = cnts > 0
                    rows_h = torch.where(some, srow[seg_lo].view(nb_p, 2), big).amin(dim=1).cpu().numpy()
                    rows_l = torch.where(some, srow[seg_hi].view(nb_p, 2), 0).amax(dim=1).cpu().numpy()
                    cnts = cnts.cpu().numpy()
                    descs.append(desc)
                    del srow
                    lo = off
                    for sgl in range(nb_p):
                        nf, nh = int(cnts[sgl][0]), int(cnts[sgl][1])
                        hi = lo + nf + nh
                        if lo >= hi:
                            continue
                        wf = NW if nh == 0 else 0 if nf == 0 else \
                            min(NW - 1, max(1, int(round(NW * nf * COST_FULL / (nf * COST_FULL + nh * COST_HALF)))))
                        tab.append((part, sgl, lo, hi, lo + nf, wf, int(rows_h[sgl]), int(rows_l[sgl])))
                        lo = hi
                    max_nb = max(max_nb, nb_p)
                    off += c
                    del row, start, idx, a, b, na, nb, full, flags, desc, wg, key, order
            if cyclic:
                # rounds of 256 workgroups: every round holds a share of EVERY tile (slot-major order), so each
                # round is one sweep over the rows by all tiles at once
                rounds = max(1, -(-n_wg // 256))
                tab.sort(key=lambda t: (t[1] % rounds, t[0], t[1]))
            blocks = torch.cat(descs).contiguous() if descs else \
                torch.zeros((0, 3 if d12 else 4), dtype=torch.int32, device=dev)
            del descs
            wg_tab = torch.tensor(tab, dtype=torch.int32, device=dev).reshape(-1, 8).contiguous()
            pb = cache[d12] = (blocks, wg_tab, max_nb)
        return pb

    @staticmethod
    def unpack_blocks(blocks: torch.Tensor) -> torch.Tensor:
        """The block list in its 16-byte form {first A entry, first B entry, row, nA | nB << 8 | flags} whichever form
        `pair_blocks` built (tests, inspection)."""
        if int(blocks.shape[1]) == 4:
            return blocks
        w = blocks[:, 2].to(torch.int64) & 0xFFFFFFFF
        meta = (((w >> 24) & 7) + 1) | ((((w >> 27) & 7) + 1) << 8) | (((w >> 30) & 3) << 16)
        return torch.stack([blocks[:, 0].to(torch.int64), blocks[:, 1].to(torch.int64), w & 0xFFFFFF, meta],
                           dim=1).to(torch.int32)

    def csc_blocks(self):
        """(rows int32, vals, bstart int64, n_blocks, col_bptr int64): the CSC form of the block with
        every column's entries cut into blocks of at most tm_cat_det_block_rows() -- the input of
        tm_csc_dense_sandwich_sorted_*.  Built once (a stable device sort by column), cached."""
        cb = getattr(self, "_cscb", None)
        if cb is None:
            from .._lib import lib

            blk = int(lib().tm_cat_det_block_rows())
            dev = self.data.device
            nnz = int(self.data.numel())
            counts = self.indptr[1:] - self.indptr[:-1]
            rows = torch.repeat_interleave(torch.arange(self.n, device=dev, dtype=torch.int32), counts)
            order = torch.sort(self.indices.to(torch.int64), stable=True).indices    # rows stay ascending
            cnt = torch.bincount(self.indices.to(torch.int64), minlength=self.m) if nnz else \
                torch.zeros(self.m, dtype=torch.int64, device=dev)
            nb = torch.div(cnt + blk - 1, blk, rounding_mode="floor")
            col_bptr = torch.zeros(self.m + 1, dtype=torch.int64, device=dev)
            torch.cumsum(nb, dim=0, out=col_bptr[1:])
            n_blocks = int(col_bptr[-1].item())
            seg0 = torch.cumsum(cnt, dim=0) - cnt
            col_of_blk = torch.repeat_interleave(torch.arange(self.m, device=dev), nb)
            within = torch.arange(n_blocks, device=dev) - col_bptr[:-1][col_of_blk]
            bstart = torch.empty(n_blocks + 1, dtype=torch.int64, device=dev)
            bstart[:-1] = seg0[col_of_blk] + within * blk
            bstart[-1] = nnz
            cb = (rows[order].contiguous(), self.data[order].contiguous(), bstart.contiguous(),
                  n_blocks, col_bptr.contiguous())
            self._cscb = cb
        return cb

    def take_rows(self, kind, *arg) -> "CsrDev":
        """Row-indexed copy built on the device: kind "slice" (lo, hi) or "index" (row ids)."""
        dev = self.data.device
        if kind == "slice":
            lo, hi = arg
            a, b = int(self.indptr[lo].item()), int(self.indptr[hi].item())
            return CsrDev(self.data[a:b].contiguous(), self.indices[a:b].contiguous(),
                          (self.indptr[lo:hi + 1] - a).contiguous(), hi - lo, self.m)
        r = D.idx_dev(arg[0], torch.int64)
        starts = self.indptr[r]
        counts = self.indptr[r + 1] - starts
        indptr = torch.zeros(r.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, dim=0, out=indptr[1:])
        nnz = int(indptr[-1].item())
        pos = torch.repeat_interleave(starts - indptr[:-1], counts) + \
            torch.arange(nnz, device=dev, dtype=torch.int64)
        return CsrDev(self.data[pos], self.indices[pos], indptr, int(r.numel()), self.m)

    @staticmethod
    def from_device_arrays(data, indices, indptr, shape) -> "CsrDev":
        """CSR arrays that already live in HBM, column indices / row pointers of EITHER width (the reference's
        `win_integral` = int32 | int64, ext/sparse.pyx:13-15): int64 column indices are narrowed and int32 row
        pointers widened ON the device (tm_index_narrow_i64 / tm_index_widen_i32); an index outside [0, m) raises."""
        from .. import _lib

        n, m = int(shape[0]), int(shape[1])
        if m >= 2**31 or n >= 2**31:
            raise ValueError("sparse block dimensions must fit int32 on the device")
        dev = data.device
        st = D.stream_ptr()
        if indices.dtype == torch.int64:
            src = indices.contiguous()
            narrow = torch.empty(src.numel(), dtype=torch.int32, device=dev)
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.call("tm_index_narrow_i64", D.p(src), src.numel(), m, D.p(narrow), D.p(bad), st)
            if int(bad.item()):
                raise ValueError("column index outside [0, m)")
            indices = narrow
        elif indices.dtype != torch.int32:
            raise TypeError("column indices must be int32 or int64")
        if indptr.dtype == torch.int32:
            src = indptr.contiguous()
            wide = torch.empty(src.numel(), dtype=torch.int64, device=dev)
            _lib.call("tm_index_widen_i32", D.p(src), src.numel(), D.p(wide), st)
            indptr = wide
        elif indptr.dtype != torch.int64:
            raise TypeError("row pointers must be int32 or int64")
        return CsrDev(data.contiguous(), indices.contiguous(), indptr.contiguous(), n, m)

    @staticmethod
    def from_scipy(csr) -> "CsrDev":
        n, m = csr.shape
        if m >= 2**31 or n >= 2**31:
            raise ValueError("sparse block dimensions must fit int32 on the device")
        return CsrDev(
            D.to_dev(np.ascontiguousarray(csr.data)),
            D.to_dev(np.ascontiguousarray(csr.indices, dtype=np.int32)),
            D.to_dev(np.ascontiguousarray(csr.indptr, dtype=np.int64)),
            n,
            m,
        )


@dataclass
class SlabCsc:
    """Slab-blocked column-major twin of a sparse block (see tm_csr_dense_sandwich_slab_* in
    include/tabmat_hip.h): rows cut into slabs of R rows, nonzeros inside a slab ordered by
    (column, row).  Built once per block from the CSR twin; the format conversion is one-off
    ingest work (a key sort) and uses torch's device sort, the products never do."""

    vals: torch.Tensor     # F[nnz]
    koff: torch.Tensor     # int32[nnz]  byte offset of the entry's row inside the LDS slab part
    cnt: torch.Tensor      # int16[S * mpad] (read as uint16) run length per (slab, column)
    gptr: torch.Tensor     # int64[S * G + 1]
    ecol: torch.Tensor     # uint8[nnz] column of the entry inside its column group
    n: int
    m: int

    @staticmethod
    def from_csr(csr: CsrDev) -> "SlabCsc":
        from .._lib import lib

        R = int(lib().tm_slab_rows())
        C = int(lib().tm_slab_group_cols())
        n, m = csr.n, csr.m
        dev = csr.data.device
        fbytes = csr.data.element_size()
        S = (n + R - 1) // R
        G = (m + C - 1) // C
        mpad = G * C
        counts = csr.indptr[1:] - csr.indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
        slab = torch.div(rows, R, rounding_mode="floor")
        key = slab * mpad + csr.indices.to(torch.int64)
        del slab
        # CSR order is (row, column)-sorted, so a STABLE sort by (slab, column) leaves the rows
        # of every (slab, column) run ascending
        key_sorted, perm = torch.sort(key, stable=True)
        del key
        vals = csr.data[perm].contiguous()
        ecol = torch.remainder(key_sorted, C).to(torch.uint8).contiguous()
        rloc = rows[perm] - torch.div(key_sorted, mpad, rounding_mode="floor") * R
        del rows, perm
        koff = (rloc * (64 * fbytes)).to(torch.int32).contiguous()
        del rloc
        cnt64 = torch.bincount(key_sorted, minlength=S * mpad) if key_sorted.numel() else \
            torch.zeros(S * mpad, dtype=torch.int64, device=dev)
        del key_sorted
        gptr = torch.zeros(S * G + 1, dtype=torch.int64, device=dev)
        if S * G:
            torch.cumsum(cnt64.view(S * G, C).sum(dim=1), dim=0, out=gptr[1:])
        cnt = cnt64.to(torch.int16).contiguous()   # <= R = 128, bit pattern == uint16
        return SlabCsc(vals, koff, cnt, gptr, ecol, n, m)


@dataclass
class SlabEll:
    """Interleaved-ELL twin of a sparse block for the static gather kernel
    (tm_csr_dense_sandwich_ell_* in include/tabmat_hip.h): rows cut into slabs of R rows, the
    columns -- sorted by density, so that the runs of a group have similar lengths -- into groups
    of C = 32; the nonzeros of one (slab, group) are I iterations of 64 slots, slot (it, c, u) =
    the (2 it + u)-th nonzero of column c, padded to the longest run of the group (the kernel skips
    the padding two columns at a time).  Built once per block (ingest: one device key sort)."""

    vals: torch.Tensor     # F[E]
    koff: torch.Tensor     # int32[E]  byte offset of the entry's row in the LDS slab, -1 = padding
    gptr: torch.Tensor     # int64[S * G + 1]  block starts (slots, multiples of 64)
    inv: torch.Tensor      # int64[m]  kernel row of column c of the block
    n: int
    m: int                 # columns of the block
    mk: int                # kernel rows = G * C
    wide: bool = False     # geometry of the wide kernel (tm_csr_dense_sandwich_ellw_*)

    @staticmethod
    def from_csr(csr: CsrDev, wide: bool = False, max_pad: float = None) -> "SlabEll":
        """wide=True: geometry of tm_csr_dense_sandwich_ellw_* (64-row slabs of 128 dense columns,
        16 columns x 4 slots per iteration) instead of tm_csr_dense_sandwich_ell_*.
        max_pad: give up (return None) when the padded stream would exceed max_pad x nnz slots --
        every non-empty (slab, group) costs at least 64 slots, so a very sparse block (<< 1 nonzero
        per slab and column group) would blow up to ~0.75 bytes per matrix cell."""
        from .._lib import lib

        R = int(lib().tm_ellw_rows() if wide else lib().tm_slab_rows())
        C = int(lib().tm_ellw_group_cols() if wide else lib().tm_slab_group_cols())
        U = 64 // C
        W = 128 if wide else 64     # dense columns per LDS slab row
        n, m = csr.n, csr.m
        dev = csr.data.device
        fbytes = csr.data.element_size()
        S = (n + R - 1) // R
        G = max(1, (m + C - 1) // C)
        mpad = G * C
        nnz = int(csr.data.numel())
        idx64 = csr.indices.to(torch.int64)
        colcnt = torch.bincount(idx64, minlength=m) if nnz else \
            torch.zeros(m, dtype=torch.int64, device=dev)
        order = torch.sort(colcnt, descending=True, stable=True).indices
        inv = torch.empty(m, dtype=torch.int64, device=dev)
        inv[order] = torch.arange(m, device=dev, dtype=torch.int64)
        del order, colcnt
        counts = csr.indptr[1:] - csr.indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
        key = torch.div(rows, R, rounding_mode="floor") * mpad + (inv[idx64] if nnz else idx64)
        del idx64
        # CSR order is row-sorted: a STABLE sort by (slab, kernel column) keeps rows ascending
        key_sorted, perm = torch.sort(key, stable=True)
        del key
        cnt64 = torch.bincount(key_sorted, minlength=S * mpad) if nnz else \
            torch.zeros(S * mpad, dtype=torch.int64, device=dev)
        iters = torch.div(cnt64.view(S * G, C).max(dim=1).values + (U - 1), U, rounding_mode="floor") \
            if S else torch.zeros(0, dtype=torch.int64, device=dev)
        gptr = torch.zeros(S * G + 1, dtype=torch.int64, device=dev)
        if S * G:
            torch.cumsum(iters * 64, dim=0, out=gptr[1:])
        total = int(gptr[-1].item())
        if max_pad is not None and total > max_pad * max(nnz, 1) and total > (1 << 22):
            return None        # too sparse for padded iterations: the caller keeps the compact stream
        vals = torch.zeros(total, dtype=csr.data.dtype, device=dev)
        koff = torch.full((total,), -1, dtype=torch.int32, device=dev)      # 0xFFFFFFFF = padding
        if nnz:
            rank = torch.arange(nnz, device=dev, dtype=torch.int64) - \
                (torch.cumsum(cnt64, dim=0) - cnt64)[key_sorted]
            del cnt64
            dst = gptr[torch.div(key_sorted, C, rounding_mode="floor")] \
                + torch.div(rank, U, rounding_mode="floor") * 64 \
                + torch.remainder(key_sorted, C) * U + torch.remainder(rank, U)
            del rank
            vals[dst] = csr.data[perm]
            rloc = rows[perm] - torch.div(key_sorted, mpad, rounding_mode="floor") * R
            koff[dst] = (rloc * (W * fbytes)).to(torch.int32)
            del rloc, dst
        return SlabEll(vals, koff, gptr, inv, n, m, mpad, wide)


@dataclass
class SlabLg:
    """Lane-group twin of a sparse block for tm_csr_dense_sandwich_lg_* (csrc/sparse_lg.hip):
    rows in slabs of R = 64, columns (sorted by density) in groups of C = 16 = one wave; column w
    of a group belongs to wave half h = w // 8 and is the half's column j = w % 8.  A round of a
    (slab, group) block is 4 chunks of 32 slots, chunk c = columns j = 2c, 2c + 1 of both halves
    with 8 positions each: slot h*16 + (j&1)*8 + it = the (8*round + it)-th nonzero of column
    8h + j as {value, koff}; koff = (1 + row in slab) * tm_lg_row_bytes(sizeof(F)), 0 = padding.  Round 0 of
    every block lies at a fixed stride.  Entries beyond a column's 8th of a slab (0.4 % of the
    columns at 5 % density) are overflow ENTRIES in xkoff: 16 bytes {value (8 bytes; float32 in
    the first 4), koff, column w = 8h + j of the group}, block by block.  The block header --
    number of entries in koff bits 20..31 of slot 0 of chunk 0, index of the first one in bits
    20..31 of slots 1..3 (3 x 12 bits) -- travels with round 0, so the kernel fetches the entries
    with SCALAR loads at the top of a slab and never waits for the vector-memory pipeline (the
    copy of the next slab) in the middle of its work.  xptr[block] (host-side view of the same
    index) is kept for tests.  Built once per block (ingest: one device key sort)."""

    vals: torch.Tensor     # F[S * G * 128]
    koff: torch.Tensor     # int32[S * G * 128]
    xptr: torch.Tensor     # int64[S * G + 1]  first overflow entry of every block
    xvals: torch.Tensor    # F[1]  (unused)
    xkoff: torch.Tensor    # int32[X * 4]  overflow entries
    inv: torch.Tensor      # int64[m]  kernel row of column c of the block
    n: int
    m: int
    mk: int                # kernel rows = G * C
    unc: int               # positions per column run without a test (2 or 4)
    # compact form (tm_csr_dense_sandwich_lgc_*), set by compact_(): vals / koff / xptr are dropped
    cvals: torch.Tensor = None   # F[real slots + 1]
    cmap: torch.Tensor = None    # uint8[S * G, 32, 4]   1 + row in slab, 0 = padding
    crec: torch.Tensor = None    # int64[S * G, 2]       {first value, entries | first entry << 32}

    @property
    def dtype(self):
        return (self.cvals if self.vals is None else self.vals).dtype

    def compact_(self) -> "SlabLg":
        """Rewrites round 0 in place as the compact stream of tm_csr_dense_sandwich_lgc_*: the
        values of the real slots only (block after block, chunk after chunk, slot order), one
        byte per slot (1 + row in slab, 0 = padding) laid out [slot][chunk] so that a lane reads
        the four chunks of its slot as one dword, and a 16-byte record per block.  At 5 % density
        40 % of the slots are real: 12 bytes per slot become 8 x 0.4 + 1 + 16 / 128."""
        from .._lib import lib

        if self.cvals is not None:
            return self
        CH, SL = 4, 32
        nblk = int(self.koff.numel()) // (CH * SL)
        rowb = int(lib().tm_lg_row_bytes(self.vals.element_size()))
        k = torch.bitwise_and(self.koff, 0xFFFFF)
        real = k != 0
        cvals = torch.cat([self.vals[real], torch.zeros(1, dtype=self.vals.dtype, device=self.vals.device)])
        cnt = real.view(nblk, CH * SL).sum(dim=1, dtype=torch.int64)
        first = torch.cumsum(cnt, dim=0) - cnt
        cmap = torch.div(k, rowb, rounding_mode="floor").to(torch.uint8).view(nblk, CH, SL) \
            .permute(0, 2, 1).contiguous()
        nrec = self.xptr[1:] - self.xptr[:-1]
        crec = torch.stack([first, nrec | (self.xptr[:-1] << 32)], dim=1).contiguous()
        self.cvals, self.cmap, self.crec = cvals.contiguous(), cmap, crec
        self.vals = self.koff = None
        return self

    @staticmethod
    def from_csr(csr: CsrDev, max_pad: float = None, max_extra: float = 0.25) -> "SlabLg":
        """Returns None when the padded stream would exceed max_pad x nnz slots (very sparse
        blocks) or more than max_extra of the blocks need further rounds (dense blocks: a column
        then holds more than 8 nonzeros per 64 rows) -- the caller keeps another twin."""
        from .._lib import lib

        R = int(lib().tm_lg_rows())
        C = int(lib().tm_lg_group_cols())
        P, W, CH, SL = 8, 128, 4, 32
        n, m = csr.n, csr.m
        dev = csr.data.device
        fbytes = csr.data.element_size()
        S = (n + R - 1) // R
        G = max(1, (m + C - 1) // C)
        mpad = G * C
        nnz = int(csr.data.numel())
        total0 = S * G * CH * SL
        if max_pad is not None and total0 > max_pad * max(nnz, 1) and total0 > (1 << 22):
            return None
        idx64 = csr.indices.to(torch.int64)
        colcnt = torch.bincount(idx64, minlength=m) if nnz else \
            torch.zeros(m, dtype=torch.int64, device=dev)
        order = torch.sort(colcnt, descending=True, stable=True).indices
        inv = torch.empty(m, dtype=torch.int64, device=dev)
        inv[order] = torch.arange(m, device=dev, dtype=torch.int64)
        del order, colcnt
        counts = csr.indptr[1:] - csr.indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
        key = torch.div(rows, R, rounding_mode="floor") * mpad + (inv[idx64] if nnz else idx64)
        del idx64
        # CSR order is row-sorted: a STABLE sort by (slab, kernel column) keeps rows ascending
        key_sorted, perm = torch.sort(key, stable=True)
        del key
        cnt64 = torch.bincount(key_sorted, minlength=S * mpad) if nnz else \
            torch.zeros(S * mpad, dtype=torch.int64, device=dev)
        rounds = torch.div(cnt64.view(S * G, C).max(dim=1).values + (P - 1), P, rounding_mode="floor") \
            if S else torch.zeros(0, dtype=torch.int64, device=dev)
        if S * G and max_extra is not None and int((rounds > 1).sum().item()) > max_extra * S * G \
                and total0 > (1 << 22):
            return None
        del rounds
        vals = torch.zeros(total0, dtype=csr.data.dtype, device=dev)
        koff = torch.zeros(total0, dtype=torch.int32, device=dev)
        xptr = torch.zeros(S * G + 1, dtype=torch.int64, device=dev)
        xvals = torch.zeros(1, dtype=csr.data.dtype, device=dev)       # (unused, kept for the ABI)
        xkoff = torch.zeros(4, dtype=torch.int32, device=dev)           # overflow entries, 4 words each
        if nnz:
            rank = torch.arange(nnz, device=dev, dtype=torch.int64) - \
                (torch.cumsum(cnt64, dim=0) - cnt64)[key_sorted]
            blk = torch.div(key_sorted, C, rounding_mode="floor")
            w = torch.remainder(key_sorted, C)
            j = torch.remainder(w, 8)
            slot = torch.div(w, 8, rounding_mode="floor") * 16 + torch.remainder(j, 2) * P \
                + torch.remainder(rank, P)
            chunk = torch.div(j, 2, rounding_mode="floor")
            rnd = torch.div(rank, P, rounding_mode="floor")
            del rank, w, j
            rloc = rows[perm] - torch.div(key_sorted, mpad, rounding_mode="floor") * R
            kv = ((rloc + 1) * int(lib().tm_lg_row_bytes(fbytes))).to(torch.int32)
            del rloc
            v = csr.data[perm]
            first = rnd == 0
            dst0 = (blk * CH + chunk) * SL + slot
            vals[dst0[first]] = v[first]
            koff[dst0[first]] = kv[first]
            later = ~first
            if bool(later.any().item()):
                # overflow ENTRIES: {value, koff, column} records of 16 bytes, block by block
                # (the entries are already sorted by block)
                nrec = torch.bincount(blk[later], minlength=S * G)
                torch.cumsum(nrec, dim=0, out=xptr[1:])
                ne = int(later.sum().item())
                xent = torch.zeros((ne, 4), dtype=torch.int32, device=dev)
                vl = v[later].contiguous()
                if fbytes == 8:
                    xent[:, 0:2] = vl.view(torch.int32).view(ne, 2)
                else:
                    xent[:, 0] = vl.view(torch.int32)
                xent[:, 2] = kv[later]
                xent[:, 3] = torch.remainder(key_sorted[later], C).to(torch.int32)   # (h << 3) | j
                xkoff = xent.reshape(-1).contiguous()
                # block header in round 0, chunk 0: slot 0 bits 20..31 = number of entries (<= 896),
                # slots 1..3 bits 20..31 = 3 x 12 bits of the first entry's index
                hdr = koff.view(S * G, CH * SL)
                hdr[:, 0] |= (nrec << 20).to(torch.int32)
                start = xptr[:-1]
                for t in range(3):
                    part = torch.bitwise_and(torch.bitwise_right_shift(start, 12 * t), 0xFFF)
                    hdr[:, 1 + t] |= torch.where(nrec > 0, part << 20, torch.zeros_like(part)).to(torch.int32)
                del nrec, xent, vl, hdr, start
            del dst0, blk, slot, chunk, rnd, v, kv, first, later
        del cnt64, key_sorted, perm, rows
        # 4 unconditional positions only pay with registers to spare (f32); measured at cfg4:
        # f64 unc=2 6.1 ms / unc=4 7.6 ms (spills), f32 5.3 / 5.3 ms
        return SlabLg(vals, koff, xptr, xvals, xkoff, inv, n, m, mpad, 2)


@dataclass
class SlabEnt:
    """Entry twin of a sparse block for tm_csr_dense_sandwich_ent_* (csrc/sparse_ent.hip, round 4): rows in
    slabs of R = 64, columns dealt to G groups of C = 16 (one wave each) in the order of their density
    (rank r -> group r % G: the groups of a workgroup carry equal shares of every slab).  The entries
    {value, (slab & 63) << 10 | row in slab << 4 | column in group} of block (group, slab) are padded to whole
    batches of 16 slots (padding: value 0, the row of the block's first entry); blocks follow one another slab
    after slab, group after group; bstart[g, s] = first batch of the block.  10 bytes per slot (f64; 12 until
    round 5: a 32-bit row << 4 | column), 1.16 slots per nonzero at 5 % density and 512 columns.  Built once per
    block on the device."""

    vals: torch.Tensor     # F[T + 192]
    meta: torch.Tensor     # int16[T + 192] (bit pattern of the 16-bit words)
    bstart: torch.Tensor   # int32[G, S + 1] (read as uint32)
    inv: torch.Tensor      # int64[m]  kernel row of column c of the block
    n: int
    m: int
    mk: int                # kernel rows = G * C

    SLACK = 192            # slots past the end that the kernel's look-ahead loads may touch

    @property
    def dtype(self):
        return self.vals.dtype

    def nbytes(self) -> int:
        return sum(int(t.numel()) * t.element_size() for t in (self.vals, self.meta, self.bstart))

    def n_slots(self) -> int:
        """Slots of the whole stream (16 x batches, the look-ahead slack not counted)."""
        return int(self.vals.numel()) - SlabEnt.SLACK

    @staticmethod
    def from_csr(csr: CsrDev, max_pad: float = None) -> "SlabEnt":
        """Returns None when the padded stream would exceed max_pad x nnz slots (very sparse blocks: every
        non-empty (group, slab) block costs at least 16 slots) or the block has 2^28 rows or more."""
        from .._lib import lib

        R = int(lib().tm_ent_rows())
        C = int(lib().tm_ent_group_cols())
        U = int(lib().tm_ent_batch_slots())
        n, m = csr.n, csr.m
        if n >= (1 << 28):
            return None
        dev = csr.data.device
        S = (n + R - 1) // R
        G = max(1, (m + C - 1) // C)
        nnz = int(csr.data.numel())
        idx64 = csr.indices.to(torch.int64)
        colcnt = torch.bincount(idx64, minlength=m) if nnz else torch.zeros(m, dtype=torch.int64, device=dev)
        order = torch.sort(colcnt, descending=True, stable=True).indices
        rank = torch.empty(m, dtype=torch.int64, device=dev)
        rank[order] = torch.arange(m, device=dev, dtype=torch.int64)
        grp_of = torch.remainder(rank, G)
        jloc_of = torch.div(rank, G, rounding_mode="floor")
        inv = grp_of * C + jloc_of
        del order, colcnt, rank
        SL = SlabEnt.SLACK
        if nnz == 0 or S == 0:
            return SlabEnt(torch.zeros(SL, dtype=csr.data.dtype, device=dev),
                           torch.zeros(SL, dtype=torch.int16, device=dev),
                           torch.zeros((G, S + 1), dtype=torch.int32, device=dev), inv, n, m, G * C)
        counts = csr.indptr[1:] - csr.indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
        key = grp_of[idx64] * S + torch.div(rows, R, rounding_mode="floor")
        cnt = torch.bincount(key, minlength=G * S)
        nb = torch.div(cnt + (U - 1), U, rounding_mode="floor")
        # round 6: the meta word carries 6 bits of the slab; an EMPTY block at every 32nd slab gets one padding batch, so
        # that two consecutive batches of a group are never 64 slabs apart (the kernels rebuild the slab from the tag and
        # a running slab).  Nothing is added where the blocks hold entries (BASELINE configs[3]: none).
        slab_of = torch.arange(G * S, device=dev, dtype=torch.int64) % S
        nb = nb + ((cnt == 0) & (slab_of % 32 == 0)).to(nb.dtype)
        total_b = int(nb.sum().item())
        if max_pad is not None and total_b * U > max_pad * nnz and total_b * U > (1 << 22):
            return None
        if total_b >= 2**31 // U:
            return None
        bst = torch.zeros(G * S + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nb, dim=0, out=bst[1:])
        # stable: CSR order is row-sorted, entries of a block stay in (row, column) order
        key_sorted, perm = torch.sort(key, stable=True)
        del key
        first = torch.cumsum(cnt, dim=0) - cnt
        pos = bst[key_sorted] * U + (torch.arange(nnz, device=dev, dtype=torch.int64) - first[key_sorted])
        del key_sorted
        T = total_b * U
        rows_p = rows[perm]
        del rows
        # meta word (16 bits): (slab & 63) << 10 | row in slab << 4 | column in group
        def word16(row, col):
            w = (((row >> 6) & 63) << 10) | ((row & 63) << 4) | col
            return torch.where(w >= 32768, w - 65536, w).to(torch.int16)

        # padding slots: value 0, the row of the block's first entry (a valid row of the same slab; an empty block
        # that holds the continuity batch: the slab's first row)
        has = nb > 0
        frow = torch.where(cnt > 0, rows_p[first.clamp(max=nnz - 1)], slab_of * R)[has]
        meta = torch.zeros(T + SL, dtype=torch.int16, device=dev)
        meta[:T] = torch.repeat_interleave(word16(frow, torch.zeros_like(frow)), nb[has] * U)
        del frow, has, first, cnt, nb, slab_of
        vals = torch.zeros(T + SL, dtype=csr.data.dtype, device=dev)
        vals[pos] = csr.data[perm]
        meta[pos] = word16(rows_p, jloc_of[idx64[perm]])
        del pos, perm, rows_p, idx64
        gi = torch.arange(G, device=dev, dtype=torch.int64)[:, None] * S + \
            torch.arange(S + 1, device=dev, dtype=torch.int64)[None, :]
        bstart = bst[gi].to(torch.int32).contiguous()
        return SlabEnt(vals, meta, bstart, inv, n, m, G * C)


def onehot_slab(cats, n: int, dtype: torch.dtype):
    """Slab form of the STACKED one-hot encodings of several categorical blocks: a sparse
    matrix with (at most) one unit entry per row and categorical, columns = the categoricals'
    columns side by side.  Lets categorical x dense cross terms run on the atomic-free gather
    kernel (csr_dense_gather_kernel) instead of LDS atomics.

    The stacked columns are PERMUTED round-robin over the kernel's column groups (32 columns per
    wave): a categorical with few levels would otherwise put all of its n nonzeros into a single
    wave.  Returns (SlabCsc over the permuted columns, inv) where row `c` of the stacked result
    is row inv[c] of the kernel output.
    cats: list of (codes int32 device tensor, n_cols, drop_first)."""
    from .._lib import lib

    C = int(lib().tm_slab_group_cols())
    dev = cats[0][0].device
    total = sum(int(c[1]) for c in cats)
    G = max(1, (total + C - 1) // C)
    # stacked column c -> permuted position: deal columns to groups like cards
    c_all = torch.arange(total, device=dev, dtype=torch.int64)
    perm_pos = (c_all % G) * C + torch.div(c_all, G, rounding_mode="floor")   # < G * C
    cols, valid = [], []
    off = 0
    for codes, ncol, drop in cats:
        c = codes.to(torch.int64) - int(bool(drop))
        ok = (c >= 0) & (c < ncol)
        cols.append(perm_pos[torch.where(ok, c + off, torch.zeros_like(c))])
        valid.append(ok)
        off += int(ncol)
    colm = torch.stack(cols, dim=1)
    okm = torch.stack(valid, dim=1)
    big = G * C
    colm = torch.where(okm, colm, torch.full_like(colm, big))
    colm, _ = torch.sort(colm, dim=1)          # CSR rows need ascending column indices
    okm = colm < big
    counts = okm.sum(dim=1)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, dim=0, out=indptr[1:])
    indices = colm[okm].to(torch.int32).contiguous()
    data = torch.ones(indices.numel(), dtype=dtype, device=dev)
    return SlabCsc.from_csr(CsrDev(data, indices, indptr, n, big)), perm_pos
