"""Row-sharded products over the GPUs of one node.

X' diag(d) X = sum_s X_s' diag(d_s) X_s over disjoint row shards, likewise X'v; X v is
row-partitioned and needs no exchange.  One process per GPU (torch.distributed, backend "nccl"
= RCCL over xGMI); each rank owns a contiguous row range of EVERY block and computes a full
p x p partial with the single-GPU kernels; the only collective on the data path is one
all-reduce of the small result (8 MB at p = 1024) per sandwich, or of a length-p vector per
transpose_matvec.  No NCCL pattern of the reference is translated: the reference has none.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced row range of `rank` (first n % world ranks get one extra row)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def bucket_rows(rows: Optional[np.ndarray], lo: int, hi: int) -> Optional[np.ndarray]:
    """Global row list -> shard-local row list (rows are unique; order is irrelevant to the
    sums).  None (= all rows) stays None."""
    if rows is None:
        return None
    rows = np.asarray(rows)
    sel = rows[(rows >= lo) & (rows < hi)] - lo
    return sel.astype(np.int32)


def shard(mat, world_size: Optional[int] = None, rank: Optional[int] = None, group=None,
          always_reduce: bool = False):
    """Row shard of `mat` (any MatrixBase, e.g. a SplitMatrix) for this rank: every block is cut
    to the rank's contiguous row range (shard_bounds) -- on the device when the block already
    lives in HBM (DenseMatrix / SparseMatrix / CategoricalMatrix.__getitem__), so a matrix built
    once can be re-partitioned without a host round trip -- and wrapped in a RowShardedMatrix.
    SURVEY.md 8e: "partitioning at construction"."""
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = mat.shape[0]
    lo, hi = shard_bounds(n, world_size, rank)
    return RowShardedMatrix(mat[lo:hi], group, bounds=(lo, hi), n_global=n, always_reduce=always_reduce)


class PendingReduce:
    """Handle of an all-reduce in flight (RowShardedMatrix.sandwich_async)."""

    def __init__(self, value, work, finish):
        self._value, self._work, self._finish = value, work, finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._value if self._finish is None else self._finish(self._value)


class RowShardedMatrix:
    """Wraps the LOCAL row shard (any MatrixBase) of a matrix that is row-partitioned over the
    ranks of `group`.  d / v passed to the products are the local slices (length = local rows);
    results of sandwich / transpose_matvec are identical on every rank after the all-reduce."""

    def __init__(self, local, group=None,
                 local_sandwich: Optional[Callable] = None,
                 local_transpose_matvec: Optional[Callable] = None,
                 bounds: Optional[tuple] = None, n_global: Optional[int] = None,
                 always_reduce: bool = False):
        self.local = local
        self.always_reduce = always_reduce   # issue the collective at world size 1 too (tests)
        self.group = group
        self.bounds = bounds            # (lo, hi) of this shard in the global row numbering
        self.n_global = n_global
        # injection points so the N>1 host logic can be exercised on CPU (gloo) in tests
        self._sandwich = local_sandwich or (lambda d, rows, cols: local.sandwich(d, rows, cols))
        self._tmv = local_transpose_matvec or (
            lambda v, rows, cols: local.transpose_matvec(v, rows, cols))
        self.shape = local.shape
        self.dtype = local.dtype

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _all_reduce(self, x):
        """Sum over the ranks.  A device result stays on the device (RCCL reduces it in place, ordered
        after the kernels on the current stream); a numpy result (numpy in -> numpy out convention)
        goes through a device buffer when the group's backend is nccl -- RCCL cannot reduce host
        memory -- and through a host tensor on gloo."""
        if not dist.is_initialized() or (self.world_size == 1 and not self.always_reduce):
            return x
        if isinstance(x, torch.Tensor):
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            return x
        t = torch.from_numpy(np.ascontiguousarray(x))
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def sandwich(self, d, rows=None, cols=None):
        """rows: LOCAL row ids of this shard (use bucket_rows for a global list)."""
        return self._all_reduce(self._sandwich(d, rows, cols))

    def sandwich_async(self, d, rows=None, cols=None) -> "PendingReduce":
        """The local partial is computed and its all-reduce STARTED (async_op): RCCL runs it on its own
        stream, ordered behind the kernels queued so far, while the caller's next launches -- typically the
        transpose_matvec of the same IRLS iteration, which does not depend on the Hessian -- proceed on the
        current stream.  `.wait()` orders the current stream behind the collective and returns the sum.
        The collective needs the COMPLETE p x p partial (every block product scatters into it), so inside one
        sandwich there is nothing to overlap it with; across the two products of an iteration there is."""
        x = self._sandwich(d, rows, cols)
        if not dist.is_initialized() or (self.world_size == 1 and not self.always_reduce):
            return PendingReduce(x, None, None)
        if isinstance(x, torch.Tensor):
            return PendingReduce(x, dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None)
        t = torch.from_numpy(np.ascontiguousarray(x))
        on_dev = dist.get_backend(self.group) == "nccl"
        if on_dev:
            t = t.cuda()
        return PendingReduce(t, dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                             (lambda y: y.cpu().numpy()) if on_dev else (lambda y: y.numpy()))

    def transpose_matvec(self, v, rows=None, cols=None):
        return self._all_reduce(self._tmv(v, rows, cols))

    def matvec(self, v, cols=None, out=None):
        """Row-partitioned output: the local rows of X v; no collective."""
        return self.local.matvec(v, cols, out)

    # ---- the same products addressed in GLOBAL row numbering (needs bounds, see shard()) ----
    def local_slice(self, x):
        """The rows of a global length-n vector (numpy or torch) that belong to this shard."""
        lo, hi = self.bounds
        return x[lo:hi]

    def sandwich_global(self, d, rows=None, cols=None):
        """d: global length-n vector; rows: global row ids or None."""
        lo, hi = self.bounds
        return self.sandwich(self.local_slice(d), bucket_rows(rows, lo, hi), cols)

    def transpose_matvec_global(self, v, rows=None, cols=None):
        lo, hi = self.bounds
        return self.transpose_matvec(self.local_slice(v), bucket_rows(rows, lo, hi), cols)
