"""Synthetic tabular designs generated directly in HBM (SURVEY.md 8d recipes), so the 10M-row
benchmark configurations never exist as 13 GB host arrays.  Used by bench.py and by the
full-size property tests; NOT part of the sandwich / matvec path itself."""
from __future__ import annotations

import numpy as np
import torch

from . import _device as D
from .categorical_matrix import CategoricalMatrix
from .dense_matrix import DenseMatrix
from .ext._types import CsrDev
from .sparse_matrix import SparseMatrix
from .split_matrix import SplitMatrix


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device=D.require_gpu())
    g.manual_seed(int(seed))
    return g


def dense_block(n: int, k: int, dtype=torch.float64, seed: int = 0, order: str = "C") -> DenseMatrix:
    g = _gen(seed)
    if order == "C":
        t = torch.randn((n, k), dtype=dtype, device=g.device, generator=g)
        return DenseMatrix(t)
    t = torch.randn((k, n), dtype=dtype, device=g.device, generator=g)
    return DenseMatrix(t.T)


def sparse_block(n: int, m: int, density: float = 0.05, dtype=torch.float64, seed: int = 0,
                 chunk: int = 1 << 20) -> SparseMatrix:
    """Bernoulli(density) pattern, U(0,1) values, built chunk-wise as a CSR twin in HBM."""
    g = _gen(seed)
    dev = g.device
    datas, inds, counts = [], [], []
    chunk = max(1, min(chunk, (1 << 30) // max(m, 1)))      # torch.nonzero: < 2^31 elements per call
    for r0 in range(0, n, chunk):
        r = min(chunk, n - r0)
        mask = torch.rand((r, m), device=dev, generator=g) < density
        counts.append(mask.sum(dim=1))
        nz = mask.nonzero(as_tuple=False)          # row-major => sorted by (row, col)
        inds.append(nz[:, 1].to(torch.int32))
        datas.append(torch.rand((nz.shape[0],), dtype=dtype, device=dev, generator=g))
        del mask, nz
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.cat(counts), dim=0, out=indptr[1:])
    csr = CsrDev(torch.cat(datas), torch.cat(inds), indptr, n, m)
    return SparseMatrix.from_device(csr)


def cat_block(n: int, n_categories: int, seed: int = 0, dtype=np.float64, drop_first=False,
              zipf: float = 0.0) -> CategoricalMatrix:
    g = _gen(seed)
    if zipf > 0:
        # inverse-CDF sampling (torch.multinomial is far too slow for 5e7 draws)
        w = 1.0 / torch.arange(1, n_categories + 1, dtype=torch.float64, device=g.device) ** zipf
        cdf = torch.cumsum(w / w.sum(), dim=0)
        u = torch.rand(n, dtype=torch.float64, device=g.device, generator=g)
        codes = torch.searchsorted(cdf, u).clamp_(max=n_categories - 1).to(torch.int32)
    else:
        codes = torch.randint(0, n_categories, (n,), dtype=torch.int32, device=g.device, generator=g)
    return CategoricalMatrix(codes, categories=np.arange(n_categories), drop_first=drop_first,
                             dtype=dtype)


def mixed_split(n: int, k_dense: int = 128, k_sparse: int = 512, cats=(256, 96, 32),
                density: float = 0.05, dtype=torch.float64, seed: int = 3) -> SplitMatrix:
    """BASELINE.json config 4: block order [dense, sparse, cat, cat, cat] => p = 1024."""
    npdt = np.float64 if dtype == torch.float64 else np.float32
    blocks = [dense_block(n, k_dense, dtype, seed), sparse_block(n, k_sparse, density, dtype, seed + 1000)]
    for i, c in enumerate(cats):
        blocks.append(cat_block(n, c, seed + 2000 + i, npdt))
    return SplitMatrix(blocks)


# The reference's own benchmark designs (src/tabmat/benchmark/generate_matrices.py:90-100):
# uniform random dense values, uniformly drawn category codes, scipy.sparse.random's default
# density of 1 %.  Same shapes and block order; generated in HBM.
REFERENCE_DESIGNS = {
    "dense": dict(n=4_000_000, dense=10),
    "sparse": dict(n=400_000, sparse=100),
    "sparse_narrow": dict(n=3_000_000, sparse=3),
    "sparse_wide": dict(n=40_000, sparse=10_000),
    "one_cat": dict(n=1_000_000, cats=(100_000,)),
    "two_cat": dict(n=1_000_000, cats=(1_000, 1_000)),
    "dense_cat": dict(n=3_000_000, cats=(1_000, 1_000), dense=5),
    "dense_smallcat": dict(n=3_000_000, cats=(10, 1_000), dense=5),
}


def reference_design(name: str, n: int | None = None, seed: int = 0):
    """One of the reference's benchmark matrices as a tabmat_amd object (n overrides the rows)."""
    spec = REFERENCE_DESIGNS[name]
    n = int(n or spec["n"])
    blocks = [cat_block(n, c, seed + 10 + i) for i, c in enumerate(spec.get("cats", ()))]
    if "dense" in spec:
        g = _gen(seed)
        blocks.append(DenseMatrix(torch.rand((n, spec["dense"]), dtype=torch.float64, device=g.device,
                                             generator=g)))
    if "sparse" in spec:
        blocks.append(sparse_block(n, spec["sparse"], 0.01, torch.float64, seed + 1))
    return blocks[0] if len(blocks) == 1 else SplitMatrix(blocks)


def algorithmic_bytes(mat) -> int:
    """Bytes of every operand read once + the output written once."""
    total = 0
    mats = mat.matrices if isinstance(mat, SplitMatrix) else [mat]
    isz = np.dtype(mat.dtype).itemsize
    n = mat.shape[0]
    for m in mats:
        if isinstance(m, DenseMatrix):
            total += m.shape[0] * m.shape[1] * isz
        elif isinstance(m, SparseMatrix):
            c = m._dev()
            total += c.data.numel() * (isz + 4) + (n + 1) * 8
        else:
            total += n * 4
    total += n * isz                       # d
    if isinstance(mat, SplitMatrix):
        total += mat.shape[1] ** 2 * 8     # float64 p x p result
    elif isinstance(mat, CategoricalMatrix):
        total += mat.shape[1] * isz        # diagonal only
    else:
        total += mat.shape[1] ** 2 * isz
    return int(total)


def algorithmic_flops(mat) -> float:
    """FMA = 2 flops, symmetric half only."""
    mats = mat.matrices if isinstance(mat, SplitMatrix) else [mat]
    n = mat.shape[0]
    fl = 0.0
    dense_k = sum(m.shape[1] for m in mats if isinstance(m, DenseMatrix))
    nnz = sum(m._dev().data.numel() for m in mats if isinstance(m, SparseMatrix))
    n_cat = sum(1 for m in mats if isinstance(m, CategoricalMatrix))
    fl += n * dense_k * (dense_k + 1)
    for m in mats:
        if isinstance(m, SparseMatrix):
            c = m._dev()
            per_row = (c.indptr[1:] - c.indptr[:-1]).to(torch.float64)
            fl += float((per_row * (per_row + 1)).sum().item())
    fl += 2.0 * nnz * dense_k + n_cat * 2.0 * n * dense_k + n_cat * 2.0 * nnz
    fl += n * (n_cat + n_cat * (n_cat - 1) / 2)
    return fl
