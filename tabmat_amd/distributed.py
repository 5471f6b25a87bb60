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


class RowShardedMatrix:
    """Wraps the LOCAL row shard (any MatrixBase) of a matrix that is row-partitioned over the
    ranks of `group`.  d / v passed to the products are the local slices (length = local rows);
    results of sandwich / transpose_matvec are identical on every rank after the all-reduce."""

    def __init__(self, local, group=None,
                 local_sandwich: Optional[Callable] = None,
                 local_transpose_matvec: Optional[Callable] = None):
        self.local = local
        self.group = group
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
        if not dist.is_initialized() or self.world_size == 1:
            return x
        if isinstance(x, torch.Tensor):
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            return x
        t = torch.from_numpy(np.ascontiguousarray(x))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def sandwich(self, d, rows=None, cols=None):
        """rows: LOCAL row ids of this shard (use bucket_rows for a global list)."""
        return self._all_reduce(self._sandwich(d, rows, cols))

    def transpose_matvec(self, v, rows=None, cols=None):
        return self._all_reduce(self._tmv(v, rows, cols))

    def matvec(self, v, cols=None, out=None):
        """Row-partitioned output: the local rows of X v; no collective."""
        return self.local.matvec(v, cols, out)
