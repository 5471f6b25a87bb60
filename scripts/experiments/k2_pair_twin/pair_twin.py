"""PairTwin builder of the K2 static-stream experiment (see README.md)."""
from dataclasses import dataclass

import torch


@dataclass
class PairTwin:
    """Pair twin of a sparse block for tm_sparse_sandwich_pair_* (csrc/sparse_pair.hip): the STATIC
    stream of the tiled self sandwich.  Rows in groups of 8, columns in chunks of 128:
    bv / bk [chunk][group][64]: slot r * 8 + t = the t-th nonzero (t < 8) of row 8 g + r inside the
    chunk as {value, column inside the chunk}, -1 = padding -- fixed stride, no pointers.  A row's
    9th, 10th ... nonzero of a chunk goes to the compact overflow arrays ov / ok =
    (row in group << 7) | column, ordered by (chunk, group, row); optr [chunk][G + 1] is the index of
    every group's first overflow entry.  Built once per block (ingest: one device key sort)."""

    bv: torch.Tensor       # F[nch * G * 64]
    bk: torch.Tensor       # int32[nch * G * 64]
    ov: torch.Tensor       # F[n_ov] (at least 1 element)
    ok: torch.Tensor       # int32[n_ov]
    optr: torch.Tensor     # int32[nch * (G + 1)]
    n: int
    m: int
    n_ov: int

    @staticmethod
    def from_csr(csr: "CsrDev", max_overflow: float = 0.25, max_pad: float = 6.0):
        """None when the form does not fit the block: more than max_overflow of the nonzeros
        beyond slot 8 of their row and chunk (dense blocks), a group with more than 64 overflow
        entries in one chunk, or a padded stream above max_pad x the nonzeros (very sparse blocks)
        -- the caller keeps the chunk-major form."""
        from .._lib import lib

        ch = int(lib().tm_sparse_chunk_cols())
        n, m = csr.n, csr.m
        dev = csr.data.device
        nnz = int(csr.data.numel())
        nch = max(1, (m + ch - 1) // ch)
        G = (n + 7) // 8
        total = nch * G * 64
        if nnz == 0 or n == 0:
            return None
        if total > max_pad * nnz and total > (1 << 22):
            return None
        counts = csr.indptr[1:] - csr.indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), counts)
        idx64 = csr.indices.to(torch.int64)
        key = torch.div(idx64, ch, rounding_mode="floor") * n + rows
        del rows, idx64
        key_sorted, perm = torch.sort(key, stable=True)     # column order inside a row is kept
        del key
        per = torch.bincount(key_sorted, minlength=nch * n)
        over = torch.clamp(per - 8, min=0)
        n_ov = int(over.sum().item())
        if n_ov > max_overflow * nnz and total > (1 << 22):
            return None
        # overflow entries per (chunk, group)
        padded = torch.zeros((nch, G * 8), dtype=torch.int64, device=dev)
        padded[:, :n] = over.view(nch, n)
        pg = padded.view(nch, G, 8).sum(dim=2)
        del padded, over
        if int(pg.max().item()) > 64 or n_ov >= 2**31:
            return None
        before = torch.cumsum(pg.reshape(-1), dim=0) - pg.reshape(-1)       # absolute, group-major
        optr = torch.empty((nch, G + 1), dtype=torch.int64, device=dev)
        optr[:, :G] = before.view(nch, G)
        optr[:-1, G] = before.view(nch, G)[1:, 0]
        optr[-1, G] = n_ov
        del pg, before
        start = torch.cumsum(per, dim=0) - per
        rank = torch.arange(nnz, device=dev, dtype=torch.int64) - start[key_sorted]
        del start, per
        c = torch.div(key_sorted, n, rounding_mode="floor")
        k = key_sorted - c * n
        del key_sorted
        col = (csr.indices[perm].to(torch.int64) - c * ch).to(torch.int32)
        vals = csr.data[perm]
        del perm
        base = rank < 8
        dst = ((c * G + torch.div(k, 8, rounding_mode="floor")) * 64 + torch.remainder(k, 8) * 8 + rank)[base]
        bv = torch.zeros(total, dtype=csr.data.dtype, device=dev)
        bk = torch.full((total,), -1, dtype=torch.int32, device=dev)
        bv[dst] = vals[base]
        bk[dst] = col[base]
        del dst
        if n_ov:
            later = ~base
            ov = vals[later].contiguous()
            ok = (torch.remainder(k[later], 8).to(torch.int32) << 7 | col[later]).contiguous()
        else:
            ov = torch.zeros(1, dtype=csr.data.dtype, device=dev)
            ok = torch.zeros(1, dtype=torch.int32, device=dev)
        return PairTwin(bv, bk, ov, ok, optr.to(torch.int32).reshape(-1).contiguous(), n, m, n_ov)


