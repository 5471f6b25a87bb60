"""Shared test inputs.  Each helper restates a fixture of the reference's test
suite as plain numpy/scipy data (block kind + arrays), so the same cases can be
fed to the CPU oracle (tests/test_oracle_*.py) and to the HIP path
(tests/test_gpu_*.py).  Citations are relative to /root/reference/.
"""
import numpy as np
from scipy import sparse as sps


def base_array(order="F"):
    """tests/test_matrices.py:13-14."""
    return np.array([[0, 0], [0, -1.0], [0, 2.0]], order=order)


def unscaled_specs():
    """tests/test_matrices.py:17-67 (get_unscaled_matrices) as (name, spec)."""
    csc = sps.csc_matrix(base_array())
    nw = base_array()
    nw.setflags(write=False)
    return [
        ("dense_F", ("dense", base_array("F"))),
        ("dense_C", ("dense", base_array("C"))),
        ("dense_ro", ("dense", nw)),
        ("sparse", ("sparse", csc)),
        ("sparse_64", ("sparse", sps.csc_matrix(
            (csc.data, csc.indices.astype(np.int64), csc.indptr.astype(np.int64)), shape=csc.shape))),
        ("cat", ("cat", np.array([1, 0, 1], dtype=np.int32), 2, False)),
        ("cat_drop", ("cat", np.array([0, 1, 2], dtype=np.int32), 3, True)),
    ]


def spec_toarray(spec, dtype=np.float64):
    kind = spec[0]
    if kind == "dense":
        return np.asarray(spec[1], dtype=dtype)
    if kind == "sparse":
        return spec[1].toarray().astype(dtype)
    codes, ncat, drop = spec[1], spec[2], spec[3]
    out = np.zeros((len(codes), max(ncat - int(drop), 0)), dtype=dtype)
    c = codes.astype(np.int64) - int(drop)
    ok = c >= 0
    out[np.nonzero(ok)[0], c[ok]] = 1
    return out


def to_oracle_block(spec):
    from oracle import oracle as orc

    kind = spec[0]
    if kind == "dense":
        return orc.Dense(spec[1])
    if kind == "sparse":
        return orc.Sparse(spec[1])
    return orc.Cat(spec[1], spec[2], spec[3])


def combine_specs(specs):
    """SplitMatrix.__init__ semantics (split_matrix.py:171-267 +
    _combine_matrices 85-141): consecutive column indices per input block, then
    all dense blocks merged into one and all sparse blocks into one (columns
    sorted by global index), categoricals untouched.  Returns (specs, indices)."""
    idx, cur = [], 0
    for s in specs:
        w = spec_toarray(s).shape[1]
        idx.append(np.arange(cur, cur + w, dtype=np.int64))
        cur += w
    out_specs, out_idx = list(specs), list(idx)
    for kind in ("dense", "sparse"):
        which = [i for i, s in enumerate(out_specs) if s[0] == kind]
        if len(which) > 1:
            new_idx = np.concatenate([out_idx[i] for i in which])
            sorter = np.argsort(new_idx)
            if kind == "dense":
                arr = np.hstack([np.asarray(out_specs[i][1]) for i in which])[:, sorter]
                merged = ("dense", np.asfortranarray(arr))
            else:
                arr = sps.hstack([out_specs[i][1] for i in which]).tocsc()[:, sorter]
                merged = ("sparse", sps.csc_matrix(arr))
            out_specs[which[0]] = merged
            out_idx[which[0]] = new_idx[sorter]
            out_specs = [s for i, s in enumerate(out_specs) if i not in which[1:]]
            out_idx = [s for i, s in enumerate(out_idx) if i not in which[1:]]
    return out_specs, out_idx


def complex_split_specs():
    """tests/test_matrices.py:70-71: SplitMatrix(get_unscaled_matrices())."""
    return combine_specs([s for _, s in unscaled_specs()])


def cat_from_values(values, missing_none=False, drop_first=False):
    """CategoricalMatrix(np.random.choice(...)) -> (codes, n_categories): sorted
    unique non-missing levels, -1 for None (categorical_matrix.py:224-230)."""
    values = np.asarray(values, dtype=object)
    mask = np.array([v is None for v in values])
    levels = np.array(sorted(set(values[~mask].tolist())))
    codes = np.full(len(values), -1, dtype=np.int32)
    codes[~mask] = np.searchsorted(levels, values[~mask].astype(levels.dtype))
    return ("cat", codes, len(levels), drop_first)


def random_split_specs(seed=0, n_rows=10, n_cols_per=3, missing=False):
    """tests/test_split_matrix.py:229-246 (random_split_matrix)."""
    if seed is not None:
        np.random.seed(seed)
    dense_1 = ("dense", np.random.random((n_rows, n_cols_per)))
    sparse = ("sparse", sps.random(n_rows, n_cols_per).tocsc())
    if missing:
        cat = cat_from_values(np.random.choice(list(range(n_cols_per)) + [None], n_rows), True)
    else:
        cat = cat_from_values(np.random.choice(range(n_cols_per), n_rows))
    dense_2 = ("dense", np.random.random((n_rows, n_cols_per)))
    cat_2 = cat_from_values(np.random.choice(range(n_cols_per), n_rows))
    return combine_specs([dense_1, sparse, cat, dense_2, cat_2])


def split_with_cat_specs(missing, idx64=False):
    """tests/test_split_matrix.py:65-107 (get_split_with_cat_components)."""
    n_rows = 10
    np.random.seed(0)
    dense_1 = ("dense", np.random.random((n_rows, 3)))
    sparse_1 = ("sparse", sps.random(n_rows, 3).tocsc())
    if missing:
        cat = cat_from_values(np.random.choice([0, 1, 2, None], n_rows), True)
    else:
        cat = cat_from_values(np.random.choice(range(3), n_rows))
    dense_2 = ("dense", np.random.random((n_rows, 3)))
    sparse_2 = ("sparse", sps.random(n_rows, 3, density=0.5).tocsc())
    c2 = cat_from_values(np.random.choice(range(3), n_rows))
    cat_2 = ("cat", c2[1], c2[2], True)
    specs, idx = combine_specs([dense_1, sparse_1, cat, dense_2, sparse_2, cat_2])
    if idx64:
        specs = [
            ("sparse", sps.csc_matrix((s[1].data, s[1].indices.astype(np.int64),
                                       s[1].indptr.astype(np.int64)), shape=s[1].shape))
            if s[0] == "sparse" else s for s in specs
        ]
    return specs, idx


def simulate_matrix(nonzero_frac=0.05, shape=(100, 50), seed=0, dtype=np.float64):
    """tests/test_fast_sandwich.py:101-110."""
    if seed is not None:
        np.random.seed(seed)
    nnz = int(np.prod(shape) * nonzero_frac)
    row_index = np.random.randint(shape[0], size=nnz)
    col_index = np.random.randint(shape[1], size=nnz)
    return sps.csr_matrix((np.random.randn(nnz).astype(dtype), (row_index, col_index)), shape)


def mixed_specs(n, k_dense, k_sparse, cats, seed, dtype=np.float64, density=0.05,
                order="C", missing=False, drop_first=False, idx_dtype=np.int32):
    """BASELINE.json cfg4-shaped mixed design, any size."""
    rng = np.random.default_rng(seed)
    specs = []
    if k_dense:
        X = rng.standard_normal((n, k_dense)).astype(dtype)
        specs.append(("dense", np.asfortranarray(X) if order == "F" else X))
    if k_sparse:
        S = sps.random(n, k_sparse, density=density, format="csc", random_state=rng,
                       dtype=np.float64).astype(dtype)
        S = sps.csc_matrix((S.data, S.indices.astype(idx_dtype), S.indptr.astype(idx_dtype)),
                           shape=S.shape)
        specs.append(("sparse", S))
    for c in cats:
        codes = rng.integers(0, c, n).astype(np.int32)
        if missing:
            codes[rng.random(n) < 0.05] = -1
        specs.append(("cat", codes, c, drop_first))
    idx, cur = [], 0
    for s in specs:
        w = spec_toarray(s).shape[1] if n <= 100000 else (
            s[1].shape[1] if s[0] != "cat" else s[2] - int(s[3]))
        idx.append(np.arange(cur, cur + w, dtype=np.int64))
        cur += w
    return specs, idx


def take_rows(spec, rows):
    """Row-indexed copy of a neutral block spec (rows: integer array, repeats allowed)."""
    rows = np.asarray(rows)
    if spec[0] == "dense":
        return ("dense", np.ascontiguousarray(spec[1][rows]))
    if spec[0] == "sparse":
        return ("sparse", spec[1].tocsr()[rows].tocsc())
    return ("cat", spec[1][rows], spec[2], spec[3])
