"""Row indexing of device-resident blocks stays in HBM (split_matrix.py:462-477,
dense_matrix.py / sparse_matrix.py / categorical_matrix.py __getitem__): the sub-matrix must equal
the host-indexed one and its products must match the oracle."""
import numpy as np
import pytest
import torch

import _cases as cs
from _gpu_util import rel_err, to_tm_split

pytestmark = pytest.mark.gpu


def _rows(kind, n, rng):
    if kind == "slice":
        return slice(n // 5, n - n // 7)
    if kind == "step":
        return slice(3, n - 1, 4)
    if kind == "array":
        return np.sort(rng.choice(n, n // 3, replace=False))
    if kind == "bool":
        return rng.random(n) < 0.4
    return [0, n - 1, 5, 5, 17]          # a list with a repeated row, unsorted


@pytest.mark.parametrize("kind", ["slice", "step", "array", "bool", "list"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_row_indexing_matches_host_indexing(kind, dtype):
    import tabmat_amd as tm
    from oracle import oracle as orc

    n = 5000
    rng = np.random.default_rng(3)
    specs, idx = cs.mixed_specs(n, 12, 40, (9, 4), seed=5)
    host = to_tm_split(specs, idx, dtype)
    dev = to_tm_split(specs, idx, dtype).to_device()
    key = _rows(kind, n, rng)
    sub_h = host[key]
    sub_d = dev[key]
    assert sub_d.shape == sub_h.shape
    for mh, md in zip(sub_h.matrices, sub_d.matrices):
        # the device path did not touch the host copy
        if isinstance(md, tm.DenseMatrix):
            assert md._array is None
        elif isinstance(md, tm.SparseMatrix):
            assert md._array is None
        else:
            assert md._host_codes is None
    np.testing.assert_array_equal(sub_d.toarray(), sub_h.toarray())
    m = sub_h.shape[0]
    d = rng.random(m).astype(dtype)
    v = rng.standard_normal(sub_h.shape[1]).astype(dtype)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    assert rel_err(sub_d.sandwich(d), sub_h.sandwich(d)) < tol
    assert rel_err(sub_d.matvec(v), sub_h.matvec(v)) < tol
    assert rel_err(sub_d.transpose_matvec(d), sub_h.transpose_matvec(d)) < tol
    # and against the oracle on the host-indexed blocks
    rows = np.arange(n)[key] if not isinstance(key, list) else np.asarray(key)
    blocks = [cs.to_oracle_block(cs.take_rows(s, rows)) for s in specs]
    want = orc.split_sandwich(blocks, idx, d.astype(np.float64))
    assert rel_err(sub_d.sandwich(d), want) < tol


def test_device_codes_are_validated():
    import tabmat_amd as tm

    codes = torch.tensor([0, 1, 5], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="exceed"):
        tm.CategoricalMatrix(codes, categories=np.arange(3))
    codes = torch.tensor([0, -1, 2], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="missing"):
        tm.CategoricalMatrix(codes, categories=np.arange(3))
    m = tm.CategoricalMatrix(codes, categories=np.arange(3), cat_missing_method="zero")
    assert m._has_missings
    m = tm.CategoricalMatrix(codes, categories=np.arange(3), cat_missing_method="convert")
    assert m.shape[1] == 4 and list(m.indices) == [0, 3, 2]


def test_mixed_dtype_operands_raise():
    import tabmat_amd as tm

    rng = np.random.default_rng(0)
    n = 6000
    cat = tm.CategoricalMatrix(rng.integers(0, 5, n))                    # nominal float64
    dense32 = tm.DenseMatrix(rng.standard_normal((n, 8)).astype(np.float32))
    d32 = rng.random(n).astype(np.float32)
    # cat x dense is templated on d / mat_j only (ext/split.pyx:32-80): float32 operands work
    got = cat._cross_sandwich(dense32, d32)
    want = cat.toarray().T @ (d32[:, None].astype(np.float64) * dense32.toarray())
    assert rel_err(got, want) < 2e-4
    with pytest.raises(TypeError):
        cat._cross_sandwich(dense32, rng.random(n))                      # float64 d, float32 B
    from scipy import sparse as sps

    with pytest.warns(UserWarning):
        split = tm.SplitMatrix([tm.DenseMatrix(rng.standard_normal((n, 3))),
                                tm.SparseMatrix(sps.random(n, 6, 0.1, format="csc", random_state=0,
                                                           dtype=np.float32))])
    with pytest.raises(TypeError):
        split.sandwich(rng.random(n))
