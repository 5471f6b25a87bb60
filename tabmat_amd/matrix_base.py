"""Abstract interface shared by every block type (reference:
/root/reference/src/tabmat/matrix_base.py:7-258).  Same method names, argument meaning and
return conventions; results are numpy arrays for numpy inputs and torch (cuda) tensors when the
per-call vector (d / v) is already a device tensor."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import numpy as np


class MatrixBase(ABC):
    """Base class for DenseMatrix, SparseMatrix, CategoricalMatrix and SplitMatrix."""

    ndim = 2
    shape: tuple
    dtype: np.dtype
    __array_priority__ = 11  # win over ndarray in `vec @ mat` (matrix_base.py:239-241)

    @abstractmethod
    def matvec(self, other, cols=None, out=None):
        """self[:, cols] @ other[cols]; `other` has full length; with `out` the product is
        ADDED into out and out is returned (matrix_base.py:15-30)."""

    @abstractmethod
    def transpose_matvec(self, vec, rows=None, cols=None, out=None):
        """self[rows, cols].T @ vec[rows]: length len(cols) without `out`;
        with `out` (length n_cols): out[cols[i]] += ... (matrix_base.py:32-61)."""

    @abstractmethod
    def sandwich(self, d, rows=None, cols=None):
        """(self[rows, cols].T * d[rows]) @ self[rows, cols] (matrix_base.py:63-76)."""

    @abstractmethod
    def getcol(self, i: int):
        ...

    @abstractmethod
    def toarray(self) -> np.ndarray:
        ...

    @abstractmethod
    def astype(self, dtype, order="K", casting="unsafe", copy=True):
        ...

    @abstractmethod
    def __getitem__(self, item):
        ...

    @abstractmethod
    def _get_col_stds(self, weights, col_means):
        ...

    @property
    def A(self) -> np.ndarray:
        return self.toarray()

    def __matmul__(self, other):
        return self.matvec(other)

    def __rmatmul__(self, other):
        """other @ X = (X.T @ other.T).T (matrix_base.py:98-114)."""
        if not hasattr(other, "T"):
            other = np.asarray(other)
        return self.transpose_matvec(other.T).T

    def _get_col_means(self, weights):
        return self.transpose_matvec(weights)

    def standardize(self, weights, center_predictors: bool, scale_predictors: bool):
        """StandardizedMatrix + column means + column stds (matrix_base.py:126-170)."""
        from .standardized_mat import StandardizedMatrix

        means = self._get_col_means(weights)
        stds = None
        mult = None
        if scale_predictors:
            stds = self._get_col_stds(weights, means)
            mult = one_over_var_inf_to_val(stds, 1.0)
        if center_predictors:
            shifter = -means * mult if mult is not None else -means
            out_means = means
        else:
            shifter = np.zeros_like(means)
            out_means = shifter
        return StandardizedMatrix(self, shifter, mult), out_means, stds

    # --- names: a thin version of matrix_base.py:176-237 -------------------------------
    def get_names(self, type: str = "column", missing_prefix: Optional[str] = None,
                  indices=None):
        names = list(getattr(self, "_colnames" if type == "column" else "_terms",
                             [None] * self.shape[1]))
        if missing_prefix is not None:
            idx = list(range(len(names))) if indices is None else indices
            names = [f"{missing_prefix}{idx[k]}" if nm is None else nm
                     for k, nm in enumerate(names)]
        return names

    def set_names(self, names, type: str = "column"):
        if isinstance(names, str):
            names = [names]
        if len(names) != self.shape[1]:
            raise ValueError(f"Length of names must be {self.shape[1]}")
        setattr(self, "_colnames" if type == "column" else "_terms", list(names))

    @property
    def column_names(self):
        return self.get_names(type="column")

    @column_names.setter
    def column_names(self, names):
        self.set_names(names, type="column")

    @property
    def term_names(self):
        return self.get_names(type="term")

    @term_names.setter
    def term_names(self, names):
        self.set_names(names, type="term")


def one_over_var_inf_to_val(arr: np.ndarray, val: float) -> np.ndarray:
    """1/arr with (near-)zeros mapped to val (matrix_base.py:244-258)."""
    arr = np.asarray(arr)
    tiny = np.abs(arr) < 1e-7
    with np.errstate(divide="ignore"):
        out = 1 / arr
    out[tiny] = val
    return out
