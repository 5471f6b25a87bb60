"""Device-resident block storage handed to the ext functions."""
from __future__ import annotations

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


@dataclass
class CsrDev:
    """CSR twin of a sparse block in HBM (sparse_matrix.py:133-143): int32 column indices,
    int64 indptr (ext/sparse.pyx:13-15 accepts int32 or int64; narrowed/widened on upload)."""

    data: torch.Tensor
    indices: torch.Tensor
    indptr: torch.Tensor
    n: int
    m: int

    @property
    def dtype(self):
        return self.data.dtype

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
