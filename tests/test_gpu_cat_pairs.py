"""All categorical x categorical tables + diagonals of a sandwich in one launch
(tm_multi_cat_pairs_*): designs with many categoricals against the oracle, including drop_first,
missing codes, row restrictions, a table too large for a bundle, float32, and counts (d = 1) that
must come out exact."""
import numpy as np
import pytest

import _cases as cs
from _gpu_util import to_tm_split

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cats,missing,drop", [((12,) * 9, False, False), ((7, 300, 40, 2, 90), True, True),
                                               ((200, 150, 3), False, True), ((5,) * 20, True, False),
                                               ((3,) * 40, True, True), ((130, 129, 127, 2), False, False),
                                               ((20_000, 5, 9, 4000), True, False), ((50,) * 7, False, True)])
def test_many_categoricals(cats, missing, drop, dtype):
    from oracle import oracle as orc

    n = 30_011
    specs, idx = cs.mixed_specs(n, 8, 0, cats, seed=len(cats), dtype=dtype, missing=missing,
                                drop_first=drop)
    X = to_tm_split(specs, idx, dtype)
    plan = X._cat_pairs_plan()
    assert plan is not None and plan.n_pairs >= len(cats)
    rng = np.random.default_rng(7)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 9)] = 0
    rows = np.sort(rng.choice(n, n // 3, replace=False))
    blocks = [cs.to_oracle_block(s) for s in specs]
    tol = 1e-10 if dtype == np.float64 else 2e-5
    for r, c in ((None, None), (rows, None), (None, np.arange(0, X.shape[1], 2))):
        got = X.sandwich(d, rows=r, cols=c)
        want = orc.split_sandwich(blocks, idx, d, r, c)
        assert np.abs(got - want).max() <= tol * np.abs(want).max()
    ones = np.ones(n, dtype=dtype)
    got = X.sandwich(ones)
    want = orc.split_sandwich(blocks, idx, ones, None, None)
    k0 = 8                                   # the categorical part holds exact counts
    assert np.array_equal(got[k0:, k0:], want[k0:, k0:])


def test_fused_equals_pairwise(monkeypatch):
    import tabmat_amd.split_matrix as smod

    specs, idx = cs.mixed_specs(20_000, 16, 60, (30, 9, 130, 4), seed=3)
    X = to_tm_split(specs, idx)
    d = np.random.default_rng(1).random(20_000)
    monkeypatch.setattr(smod, "CAT_PAIRS_FUSED", False)
    want = X.sandwich(d)
    monkeypatch.setattr(smod, "CAT_PAIRS_FUSED", True)
    got = X.sandwich(d)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cats,missing,drop", [((12,) * 9, False, False), ((7, 300, 40, 2, 90), True, True),
                                               ((3,) * 70, True, False), ((20_000, 5, 9), True, True),
                                               ((50, 60), False, False)])
def test_fused_matvec_and_transpose_matvec(cats, missing, drop, dtype):
    """SplitMatrix.matvec / transpose_matvec with all categorical blocks in one launch each
    (tm_multi_cat_matvec_*, the diagonals-only form of tm_multi_cat_pairs_*) against the oracle."""
    import torch
    from oracle import oracle as orc

    n = 25_013
    specs, idx = cs.mixed_specs(n, 6, 20, cats, seed=len(cats) + 1, dtype=dtype, missing=missing,
                                drop_first=drop)
    X = to_tm_split(specs, idx, dtype)
    assert X._cat_hist_plan() is not None
    blocks = [cs.to_oracle_block(s) for s in specs]
    rng = np.random.default_rng(11)
    p = X.shape[1]
    v = rng.standard_normal(p).astype(dtype)
    w = rng.standard_normal(n).astype(dtype)
    w[rng.integers(0, n, n // 7)] = 0
    rows = np.sort(rng.choice(n, n // 4, replace=False))
    cols = np.arange(1, p, 3)
    tol = 1e-11 if dtype == np.float64 else 3e-5
    for c in (None, cols):
        got = X.matvec(v, cols=c)
        want = orc.split_matvec(blocks, idx, v, c)
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    for r, c in ((None, None), (rows, None), (None, cols), (rows, cols), ([], None)):
        got = X.transpose_matvec(w, rows=r, cols=c)
        want = orc.split_transpose_matvec(blocks, idx, w, r, c)
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    # device operands, out= accumulation
    vd = torch.as_tensor(v, device="cuda")
    out = torch.ones(n, dtype=vd.dtype, device="cuda")
    X.matvec(vd, out=out)
    assert np.abs(out.cpu().numpy() - 1 - orc.split_matvec(blocks, idx, v, None)).max() <= \
        tol * max(1.0, np.abs(v).max() * len(cats))
    ones = np.ones(n, dtype=dtype)          # counts come out exact
    got = X.transpose_matvec(ones)
    want = orc.split_transpose_matvec(blocks, idx, ones, None, None)
    assert np.array_equal(got[26:], want[26:])
