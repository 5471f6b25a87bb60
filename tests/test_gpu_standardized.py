"""StandardizedMatrix over device blocks: the reference's tests/test_standardized_mat.py restated
(same seed, same matrices, same assertions), the device-vector path (torch in -> torch out), row /
column restrictions against dense algebra, and the one-pass property of the sandwich (X' d comes
out of the sandwich pass of a split with a complete categorical: no transpose_matvec kernels)."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs
from _gpu_util import rel_err, to_tm_block, to_tm_split

pytestmark = pytest.mark.gpu


def _reference_case():
    import tabmat_amd as tm

    np.random.seed(0)
    n_rows, n_cols = 8, 5
    sp_mat = tm.SparseMatrix(sps.random(n_rows, n_cols, density=0.8))
    col_shift = np.random.uniform(0, 1, n_cols)
    col_mult = np.random.uniform(0.5, 1.5, n_cols)
    expected = col_mult[None, :] * sp_mat.toarray() + col_shift[None, :]
    return tm.StandardizedMatrix(sp_mat, col_shift, col_mult), expected


def test_setup_and_densify_col():
    std, expected = _reference_case()
    assert std.toarray().shape == (8, 5)
    np.testing.assert_almost_equal(std.toarray(), expected)


def test_standardized_matvec():
    std, expected = _reference_case()
    v = np.random.rand(std.shape[1])
    np.testing.assert_almost_equal(std.matvec(v), expected.dot(v))
    np.testing.assert_almost_equal(std.matvec(torch.from_numpy(v).cuda()).cpu().numpy(), expected.dot(v))


def test_standardized_transpose_matvec():
    std, expected = _reference_case()
    v = np.random.rand(std.shape[0])
    np.testing.assert_almost_equal(std.transpose_matvec(v), v @ expected)
    np.testing.assert_almost_equal(std.transpose_matvec(torch.from_numpy(v).cuda()).cpu().numpy(),
                                   v @ expected)


def test_standardized_sandwich():
    std, expected = _reference_case()
    v = np.random.rand(std.shape[0])
    want = (expected.T * v) @ expected
    np.testing.assert_almost_equal(std.sandwich(v), want)
    got = std.sandwich(torch.from_numpy(v).cuda())
    assert isinstance(got, torch.Tensor) and got.is_cuda
    np.testing.assert_almost_equal(got.cpu().numpy(), want)


def test_zero_sd_cols():
    import tabmat_amd as tm

    n_rows = 100
    weights = np.ones(n_rows) / n_rows
    X = tm.DenseMatrix(np.ones([n_rows, 1])).standardize(weights, True, True)[0]
    assert X.mult == 1


def _blocks(kind, n, rng):
    import tabmat_amd as tm

    if kind == "dense":
        return tm.DenseMatrix(rng.standard_normal((n, 7)))
    if kind == "sparse":
        return tm.SparseMatrix(sps.random(n, 9, 0.2, format="csc", random_state=3))
    if kind == "cat":
        return tm.CategoricalMatrix(rng.integers(0, 6, n))
    if kind == "cat_drop_missing":
        c = rng.integers(-1, 6, n)
        return tm.CategoricalMatrix(c, categories=np.arange(6), drop_first=True, cat_missing_method="zero")
    specs, idx = cs.mixed_specs(n, 6, 9, (4, 3), seed=4)
    return to_tm_split(specs, idx)


@pytest.mark.parametrize("kind", ["dense", "sparse", "cat", "cat_drop_missing", "split"])
@pytest.mark.parametrize("use_mult", [True, False])
@pytest.mark.parametrize("restrict", [False, True])
def test_standardized_products_all_block_kinds(kind, use_mult, restrict):
    import tabmat_amd as tm

    rng = np.random.default_rng(11)
    n = 3000
    mat = _blocks(kind, n, rng)
    p = mat.shape[1]
    shift = rng.standard_normal(p)
    mult = rng.uniform(0.5, 1.5, p) if use_mult else None
    std = tm.StandardizedMatrix(mat, shift, mult)
    S = (mat.toarray() * (mult[None, :] if use_mult else 1.0)) + shift[None, :]
    rows = np.sort(rng.choice(n, n // 2, replace=False)) if restrict else None
    cols = np.sort(rng.choice(p, max(1, p // 2), replace=False)) if restrict else None
    Sr = S[rows if rows is not None else slice(None)][:, cols if cols is not None else slice(None)]
    d = rng.random(n)
    dr = d[rows] if rows is not None else d
    want = (Sr.T * dr) @ Sr
    scale = np.abs(want).max()
    for dd in (d, torch.from_numpy(d).cuda()):
        got = std.sandwich(dd, rows, cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        assert got.dtype == np.float64
        assert np.abs(got - want).max() / scale < 1e-10
    want_t = Sr.T @ dr
    for dd in (d, torch.from_numpy(d).cuda()):
        got = std.transpose_matvec(dd, rows, cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        assert np.abs(got - want_t).max() / np.abs(want_t).max() < 1e-10
    v = rng.standard_normal(p)
    Sc = S[:, cols] if cols is not None else S
    want_m = Sc @ (v[cols] if cols is not None else v)
    for vv in (v, torch.from_numpy(v).cuda()):
        got = std.matvec(vv, cols)
        got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
        assert np.abs(got - want_m).max() / np.abs(want_m).max() < 1e-10


def test_split_sandwich_yields_xtd_without_a_second_pass():
    """A split with a complete categorical: X' d of every block falls out of the sandwich pass;
    the separate transpose_matvec entry points are never called."""
    import tabmat_amd as tm
    from tabmat_amd import _lib

    specs, idx = cs.mixed_specs(20_000, 16, 40, (8, 5), seed=2)
    mat = to_tm_split(specs, idx)
    assert any(isinstance(m, tm.CategoricalMatrix) and not m.drop_first and not m._has_missings
               for m in mat.matrices)
    d = torch.rand(mat.shape[0], dtype=torch.float64, device="cuda")
    seen = []
    real_call = _lib.call

    def spy(name, *a):
        seen.append(name)
        return real_call(name, *a)

    import tabmat_amd.ext.categorical as xc
    import tabmat_amd.ext.dense as xd
    import tabmat_amd.ext.sparse as xs
    import tabmat_amd.ext.split as xsp

    mods = [xc, xd, xs, xsp]
    old = [m.call for m in mods]
    for m in mods:
        m.call = spy
    try:
        inner, xtd = mat._sandwich_xtd_dev(d, None, None)
    finally:
        for m, o in zip(mods, old):
            m.call = o
    assert not any("rmatvec" in s or "matvec" in s and "transpose" not in s for s in seen), seen
    # categorical self terms use tm_cat_transpose_matvec_* for their DIAGONAL (that is the sandwich
    # kernel of a categorical block); no dense / sparse transpose-matvec entry point was needed
    assert not any(s.startswith("tm_dense_rmatvec") or s.startswith("tm_csr_rmatvec") for s in seen)
    want = mat.transpose_matvec(d)
    assert float((xtd - want).abs().max() / want.abs().max()) < 1e-12
    full = mat.sandwich(d)
    # (LDS-atomic kernels sum in run-dependent order: equal up to rounding, not bitwise)
    assert float((inner - full).abs().max() / full.abs().max()) < 1e-13


def _spy_xtd(mat, d):
    """(inner, xtd, names of the C-ABI entry points called) of one _sandwich_xtd_dev."""
    from tabmat_amd import _lib

    import tabmat_amd.ext.categorical as xc
    import tabmat_amd.ext.dense as xd
    import tabmat_amd.ext.sparse as xs
    import tabmat_amd.ext.split as xsp

    seen = []
    real_call = _lib.call

    def spy(name, *a):
        seen.append(name)
        return real_call(name, *a)

    mods = [xc, xd, xs, xsp]
    old = [m.call for m in mods]
    for m in mods:
        m.call = spy
    try:
        inner, xtd = mat._sandwich_xtd_dev(d, None, None)
    finally:
        for m, o in zip(mods, old):
            m.call = o
    return inner, xtd, seen


@pytest.mark.parametrize("design", ["drop_first_everywhere", "no_categorical"])
def test_xtd_in_one_pass_without_a_complete_categorical(design):
    """X' d of the dense block comes out of the syrk's own pass
    (tm_dense_sandwich_i8_xtd_f64 / tm_dense_sandwich_co_f64) and X' d of the sparse block out of the gather's stream loop
    (tm_csr_dense_sandwich_ent_* with a colsum pointer since round 4; tm_csr_dense_sandwich_lg_xtd_* / lgc
    before): no transpose_matvec entry point runs even when no categorical
    block is complete (reference: standardized_mat.py:149-150 makes a second pass)."""
    import tabmat_amd as tm

    n = 30_000
    specs, idx = cs.mixed_specs(n, 96, 160, (9, 6), seed=4)
    blocks = []
    for sp in specs:
        if sp[0] == "cat":
            if design == "no_categorical":
                continue
            sp = (sp[0], sp[1], sp[2], True)          # drop_first on every categorical
        blocks.append(to_tm_block(sp))
    mat = tm.SplitMatrix(blocks)
    assert not any(isinstance(m, tm.CategoricalMatrix) and not m.drop_first for m in mat.matrices)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    inner, xtd, seen = _spy_xtd(mat, d)
    assert not any(s.startswith(("tm_dense_rmatvec", "tm_csr_rmatvec", "tm_dense_matvec",
                                 "tm_csr_matvec")) for s in seen), seen
    assert any(s.startswith("tm_dense_sandwich_i8_") or s == "tm_dense_sandwich_co_f64" for s in seen)
    assert any(s.startswith(("tm_csr_dense_sandwich_ent_", "tm_csr_dense_sandwich_lgc_",
                             "tm_csr_dense_sandwich_lg_xtd")) for s in seen)
    want = mat.transpose_matvec(d)
    assert float((xtd - want).abs().max() / want.abs().max()) < 1e-12
    full = mat.sandwich(d)
    # (mat.sandwich takes the int8-sliced syrk for the dense block, the X'd form the f64 syrk)
    assert float((inner - full).abs().max() / full.abs().max()) < 1e-12
    # and the standardized product built on it matches dense algebra
    Xh = np.empty(mat.shape)
    for m, ix in zip(mat.matrices, mat.indices):
        Xh[:, ix] = m.toarray()
    w = np.full(n, 1.0 / n)
    std, means, stds = mat.standardize(w, True, True)
    S = (Xh - means) / stds
    dh = d.cpu().numpy()
    got = std.sandwich(d).cpu().numpy()
    want_s = (S.T * dh) @ S
    assert np.abs(got - want_s).max() / np.abs(want_s).max() < 1e-10


def test_k3_column_sums_vs_oracle():
    """tm_csr_dense_sandwich_lg_xtd_*: both outputs against the oracle, f64 and f32, with overflow
    entries (dense columns) and zero weights."""
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(9)
    n, m, k = 9_001, 70, 128
    dens = np.concatenate([np.full(60, 0.04), np.full(10, 0.3)])      # 10 columns overflow the round
    mask = rng.random((n, m)) < dens
    Sd = np.where(mask, rng.random((n, m)), 0.0)
    B = rng.standard_normal((n, k))
    d = rng.random(n)
    d[::7] = 0.0
    for dt, tol in ((np.float64, 1e-10), (np.float32, 2e-5)):
        A = tm_sparse(Sd.astype(dt))
        Bd = tm_dense(B.astype(dt))
        lg = A._lg()
        assert lg is not None
        out, csum = xs.csr_dense_sandwich_lg(lg, Bd._dev_c(), torch.from_numpy(d.astype(dt)).cuda(),
                                             want_colsum=True)
        want = (Sd.T * d) @ B
        assert rel_err(out.cpu().numpy(), want) < tol
        assert rel_err(csum.cpu().numpy(), Sd.T @ d) < tol


def tm_sparse(a):
    import tabmat_amd as tm

    return tm.SparseMatrix(sps.csc_matrix(a))


def tm_dense(a):
    import tabmat_amd as tm

    return tm.DenseMatrix(a)
