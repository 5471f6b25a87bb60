"""Sparse blocks beyond the 32-bit entry index of the twins are worked on in row parts (the
sandwich is a sum over rows).  The real limit is 2^31 nonzeros; the tests force the split with a
small PART_NNZ and compare with the oracle."""
import numpy as np
import pytest
from scipy import sparse as sps

import _cases as cs
from _gpu_util import to_tm_split

pytestmark = pytest.mark.gpu


def test_split_matrix_in_row_parts(monkeypatch):
    import tabmat_amd.sparse_matrix as spm
    from oracle import oracle as orc

    monkeypatch.setattr(spm, "PART_NNZ", 30_000)
    specs, idx = cs.mixed_specs(12_000, 48, 160, (40, 700), seed=21)      # ~96k nonzeros: 4+ parts
    X = to_tm_split(specs, idx)
    parts = X._parts()
    assert parts is not None and len(parts) >= 4 and parts[0][0] == 0 and parts[-1][1] == 12_000
    rng = np.random.default_rng(3)
    d = rng.random(12_000)
    rows = np.sort(rng.choice(12_000, 5000, replace=False))
    cols = np.arange(1, X.shape[1], 3)
    blocks = [cs.to_oracle_block(s) for s in specs]
    for r, c in ((None, None), (rows, None), (rows[:40], cols)):
        got = X.sandwich(d, rows=r, cols=c)
        want = orc.split_sandwich(blocks, idx, d, r, c)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    # matvec / transpose_matvec do not depend on the parts (64-bit row pointers)
    v = rng.standard_normal(X.shape[1])
    E = np.hstack([cs.spec_toarray(s) for s in specs])
    assert np.abs(X.matvec(v) - E @ v).max() <= 1e-10 * np.abs(E @ v).max()


def test_standardized_sandwich_in_row_parts(monkeypatch):
    import tabmat_amd as tm
    import tabmat_amd.sparse_matrix as spm

    monkeypatch.setattr(spm, "PART_NNZ", 20_000)
    specs, idx = cs.mixed_specs(9000, 16, 100, (12,), seed=2)
    X = to_tm_split(specs, idx)
    E = np.hstack([cs.spec_toarray(s) for s in specs])
    rng = np.random.default_rng(1)
    shift, mult = rng.standard_normal(E.shape[1]), rng.random(E.shape[1]) + 0.5
    S = tm.StandardizedMatrix(X, shift, mult)
    d = rng.random(9000)
    Es = E * mult + shift
    want = Es.T @ (d[:, None] * Es)
    assert np.abs(S.sandwich(d) - want).max() <= 1e-10 * np.abs(want).max()


def test_sparse_matrix_in_row_parts(monkeypatch):
    import tabmat_amd as tm
    import tabmat_amd.sparse_matrix as spm

    monkeypatch.setattr(spm, "PART_NNZ", 10_000)
    rng = np.random.default_rng(5)
    S = sps.random(8000, 300, density=0.02, format="csc", random_state=rng)
    d = rng.random(8000)
    sm = tm.SparseMatrix(S)
    want = (S.T @ sps.diags(d) @ S).toarray()
    assert np.abs(sm.sandwich(d) - want).max() <= 1e-10 * np.abs(want).max()
    rows = np.sort(rng.choice(8000, 900, replace=False))
    Sr = S.tocsr()[rows]
    want = (Sr.T @ sps.diags(d[rows]) @ Sr).toarray()
    assert np.abs(sm.sandwich(d, rows=rows) - want).max() <= 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("n_sel", [30, 70, 100])
def test_standardized_sandwich_in_row_parts_with_a_narrow_column_selection(monkeypatch, n_sel):
    """the row-parts path with a narrow `cols=` selection that mixes dense and sparse columns.
    The narrow path marks EVERY block of the selection as centred (sparse columns and all-zero centres have centre
    0); the parts' column sums must take those as they are instead of asking for a centre vector."""
    import tabmat_amd as tm
    import tabmat_amd.sparse_matrix as spm

    monkeypatch.setattr(spm, "PART_NNZ", 20_000)
    specs, idx = cs.mixed_specs(9000, 64, 120, (12,), seed=4)
    X = to_tm_split(specs, idx)
    assert X._parts() is not None and len(X._parts()) >= 2
    E = np.hstack([cs.spec_toarray(s) for s in specs])
    order = np.argsort(np.concatenate(idx))         # E's columns in block order -> matrix column order
    E = E[:, order]
    rng = np.random.default_rng(n_sel)
    p = E.shape[1]
    shift, mult = rng.standard_normal(p) * 3.0, rng.random(p) + 0.5
    S = tm.StandardizedMatrix(X, shift, mult)
    d = rng.random(9000)
    noncat = np.concatenate([idx[0], idx[1]])
    cols = np.sort(np.concatenate([rng.choice(noncat, n_sel, replace=False), idx[2][:5]]))
    Es = (E * mult + shift)[:, cols]
    want = Es.T @ (d[:, None] * Es)
    got = S.sandwich(d, cols=cols)
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
