"""StandardizedMatrix.sandwich on UNCENTRED columns: dense columns whose mean is
10, 400 and 1e4 standard deviations away from zero ("age"-, "year"- / "price"- and id-like columns), >= 30k
rows so that the default int8-sliced dense term runs (I8_MIN_ROWS = 4096), float64.  The expected values are
LONG-DOUBLE dense algebra on the standardized matrix  Z = mult * X + shift  (the acceptance rule of the
reference's tests/test_real_matrix.py:17-33, which compares X_std.sandwich with the dense sandwich of
X_std.toarray(), taken at 64-bit mantissa instead of 53), and the bar is the north star's 1e-10, asserted
ENTRY BY ENTRY at the natural scale sqrt(S_ii S_jj) of the STANDARDIZED result (_gpu_util.nat_err).

Why this needs its own kernels' support: the reference forms the raw product X' D X and subtracts mean-sized
rank-one terms (standardized_mat.py:148-171), which amplifies the error of the product by (mean / std)^2.  The
dense kernels here take the column centres and compute the product of X - 1 c' (tm_dense_sandwich_*_centered_*).
"""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

from _gpu_util import nat_err

pytestmark = pytest.mark.gpu

LD = np.longdouble
BAR = 1e-10
RATIOS = (0.0, 10.0, 400.0, 1e4)


def _dense_cols(rng, n, k, order="C"):
    """k columns mean_j + std_j z with mean / std cycling through RATIOS and std through (1, 5, 0.02, 300)."""
    stds = np.array([(1.0, 5.0, 0.02, 300.0)[(j // len(RATIOS)) % 4] for j in range(k)])
    means = np.array([RATIOS[j % len(RATIOS)] for j in range(k)]) * stds
    means[1::8] *= -1.0                         # a few negative means
    X = means[None, :] + stds[None, :] * rng.standard_normal((n, k))
    return np.asfortranarray(X) if order == "F" else np.ascontiguousarray(X)


def _ld_sandwich(Z, d, rows=None):
    """Z' diag(d) Z in long double (Z: long double, row chunks keep the temporaries small)."""
    if rows is not None:
        Z, d = Z[rows], d[rows]
    p = Z.shape[1]
    S = np.zeros((p, p), dtype=LD)
    dl = d.astype(LD)
    for a in range(0, Z.shape[0], 8192):
        Zc = Z[a:a + 8192]
        S += (Zc.T * dl[a:a + 8192]) @ Zc
    return S


def _standardized_ld(std):
    """The standardized matrix in long double from the float64 shift / mult the object holds."""
    X = std.mat.toarray().astype(LD)
    mult = np.ones(std.shape[1], dtype=LD) if std.mult is None else std.mult.astype(LD)
    return X * mult[None, :] + std.shift.astype(LD)[None, :]


def _check(got, want_ld, what):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.dtype == np.float64
    want = np.asarray(want_ld, dtype=np.float64)
    # (the float64 image of the long-double result: its own rounding is 1e-16 of each entry)
    err = nat_err(got, want)
    assert err < BAR, f"{what}: entry-wise error {err:.2e} at the natural scale (bar {BAR:.0e})"
    return err


@pytest.fixture(scope="module")
def split_case():
    import tabmat_amd as tm

    rng = np.random.default_rng(50)
    n = 32768
    Xd = _dense_cols(rng, n, 72)
    Xs = sps.random(n, 24, density=0.05, format="csc", random_state=rng)
    c1 = rng.integers(0, 20, n)
    c2 = rng.integers(0, 7, n)
    mat = tm.SplitMatrix([tm.DenseMatrix(Xd), tm.SparseMatrix(Xs), tm.CategoricalMatrix(c1),
                          tm.CategoricalMatrix(c2, drop_first=True)])
    w = rng.random(n)
    w /= w.sum()
    std = mat.standardize(w, True, True)[0]
    d = rng.random(n)
    Z = _standardized_ld(std)
    return dict(std=std, d=d, Z=Z, full=_ld_sandwich(Z, d), n=n, p=mat.shape[1], rng=rng)


def test_split_standardized_sandwich_uncentred_columns(split_case):
    c = split_case
    ratios = np.abs(c["std"].shift[:72])            # |shift| = |mean| / std of the dense columns
    assert ratios.max() > 5e3 and (ratios > 300).sum() >= 18, "the fixture must hold year- and id-like columns"
    for dd in (c["d"], torch.from_numpy(c["d"]).cuda()):
        _check(c["std"].sandwich(dd), c["full"], "all rows, all columns")


def test_split_standardized_sandwich_rows(split_case):
    c = split_case
    rng = np.random.default_rng(51)
    for share in (0.5, 0.08):                         # masked-d pass of the int8 kernel / row-list kernels
        rows = np.sort(rng.choice(c["n"], int(share * c["n"]), replace=False))
        want = _ld_sandwich(c["Z"], c["d"], rows)
        _check(c["std"].sandwich(c["d"], rows=rows), want, f"{share:.0%} of the rows")


def test_split_standardized_sandwich_cols(split_case):
    c = split_case
    rng = np.random.default_rng(52)
    p = c["p"]
    for k in (int(0.7 * p), 40, 9):                   # full product + selection / generic / one dense block
        cols = np.sort(rng.choice(p, k, replace=False))
        got = c["std"].sandwich(c["d"], cols=cols)
        _check(got, c["full"][np.ix_(cols, cols)], f"{k} of {p} columns")
    rows = np.sort(rng.choice(c["n"], c["n"] // 3, replace=False))
    cols = np.sort(rng.choice(p, 30, replace=False))
    want = _ld_sandwich(c["Z"][:, cols], c["d"], rows)
    _check(c["std"].sandwich(torch.from_numpy(c["d"]).cuda(), rows=rows, cols=cols), want, "rows and columns")


def test_split_standardized_sandwich_strict_f64(split_case):
    """The float64 kernels (K1c / generic syrk) take the centres too."""
    import tabmat_amd as tm
    from tabmat_amd import dense_matrix as dm

    c = split_case
    old = dm.set_strict_f64(True)
    try:
        _check(c["std"].sandwich(c["d"]), c["full"], "strict f64")
    finally:
        dm.set_strict_f64(old)


def test_centring_is_what_closes_the_gap(split_case):
    """Without the centred kernels (the reference's formula on the raw product) the same call misses the bar on
    the id-like columns by orders of magnitude: the test above would have caught round 4's default path."""
    c = split_case
    std = c["std"]
    err_c = nat_err(std.sandwich(c["d"]), np.asarray(c["full"], dtype=np.float64))
    raw = type(std)(std.mat, std.shift, std.mult)
    raw.CENTER_DENSE = False
    err_r = nat_err(raw.sandwich(c["d"]), np.asarray(c["full"], dtype=np.float64))
    assert err_c < BAR
    assert err_r > 100 * err_c, (err_r, err_c)


@pytest.mark.parametrize("variant", ["i8_128", "odd_71", "f_order_80", "small_n", "wide_200", "no_mult", "narrow_9"])
def test_dense_standardized_sandwich_variants(variant):
    """DenseMatrix inside a StandardizedMatrix: every dense syrk takes the centres -- K1e (int8, <= 128 even
    columns, with X' d from the same pass), the generic MFMA syrk (odd widths, F order without / with the
    row-major twin, fewer rows than the int8 kernel wants), the 128-column panels of a wide block."""
    import tabmat_amd as tm

    rng = np.random.default_rng(60)
    n, k, order = {"i8_128": (30016, 128, "C"), "odd_71": (30000, 71, "C"), "f_order_80": (30000, 80, "F"),
                   "small_n": (3000, 96, "C"), "wide_200": (8192, 200, "C"), "no_mult": (30000, 66, "C"),
                   "narrow_9": (30000, 9, "C")}[variant]
    X = _dense_cols(rng, n, k, order)
    mat = tm.DenseMatrix(X)
    w = np.full(n, 1.0 / n)
    std = mat.standardize(w, True, variant != "no_mult")[0]
    d = rng.random(n)
    Z = _standardized_ld(std)
    want = _ld_sandwich(Z, d)
    if variant == "no_mult":
        # centred, unscaled columns: the natural scale of an entry is std_i std_j sum(d), still far below mean^2
        assert std.mult is None
    _check(std.sandwich(d), want, variant)
    _check(std.sandwich(torch.from_numpy(d).cuda()), want, variant + " (device vector)")
    rows = np.sort(rng.choice(n, n // 2, replace=False))
    cols = np.sort(rng.choice(k, max(2, k // 3), replace=False))
    _check(std.sandwich(d, rows=rows, cols=cols), _ld_sandwich(Z[:, cols], d, rows), variant + " rows / cols")


def test_centred_c_abi_entry_points_match_explicit_centring():
    """tm_dense_sandwich{,_co,_i8}_centered_f64 against the plain entry points run on an explicitly centred copy
    of the block (the same kernels, so the two agree to the rounding of x - c, which is exact here: the centre
    and the entries share their exponent range), and the centred column sums."""
    from tabmat_amd.ext import dense as xd
    from tabmat_amd.ext._types import DenseDev

    rng = np.random.default_rng(61)
    n, k = 20000, 96
    X = _dense_cols(rng, n, k)
    c = np.round(X.mean(axis=0), 3)
    Xc = X - c[None, :]
    d = rng.random(n)
    dev = DenseDev.from_host(X)
    devc = DenseDev.from_host(Xc)
    dd = torch.from_numpy(d).cuda()
    cd = torch.from_numpy(c).cuda()
    want = (Xc.T.astype(LD) * d.astype(LD)) @ Xc.astype(LD)
    want_cs = (Xc.T.astype(LD) @ d.astype(LD)).astype(np.float64)
    g = xd.dense_sandwich(dev, dd, None, None, center=cd)
    assert nat_err(g.cpu().numpy(), np.asarray(want, dtype=np.float64)) < 1e-13
    g2, cs2 = xd.dense_sandwich_co(dev, dd, want_colsum=True, center=cd)
    assert nat_err(g2.cpu().numpy(), np.asarray(want, dtype=np.float64)) < 1e-13
    np.testing.assert_allclose(cs2.cpu().numpy(), want_cs, rtol=0, atol=1e-12 * np.abs(Xc).max() * d.sum())
    cmax = torch.from_numpy(np.abs(Xc).max(axis=0)).cuda()
    g3, cs3 = xd.dense_sandwich_i8(dev, dd, cmax, want_colsum=True, center=cd)
    assert nat_err(g3.cpu().numpy(), np.asarray(want, dtype=np.float64)) < 1e-12
    np.testing.assert_allclose(cs3.cpu().numpy(), want_cs, rtol=0, atol=1e-12 * np.abs(Xc).max() * d.sum())
    # the int8 kernel on the explicitly centred copy: same digits
    g4 = xd.dense_sandwich_i8(devc, dd, cmax)
    assert nat_err(g3.cpu().numpy(), g4.cpu().numpy()) < 1e-13
    rows = torch.from_numpy(np.sort(rng.choice(n, 700, replace=False)).astype(np.int32)).cuda()
    cols = torch.from_numpy(np.sort(rng.choice(k, 17, replace=False)).astype(np.int32)).cuda()
    g5 = xd.dense_sandwich(dev, dd, rows, cols, center=cd)
    r, cc = rows.cpu().numpy(), cols.cpu().numpy()
    w5 = (Xc[np.ix_(r, cc)].T * d[r]) @ Xc[np.ix_(r, cc)]
    assert nat_err(g5.cpu().numpy(), w5) < 1e-13


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("order,n,k", [("F", 6000, 80), ("F", 6001, 37), ("C", 5000, 300), ("F", 4000, 200),
                                       ("C", 3000, 18), ("C", 7000, 131)])
def test_generic_centred_syrk_every_load_mode(dtype, order, n, k, monkeypatch):
    """tm_dense_sandwich_centered_{f32,f64} through every load mode of the generic MFMA syrk: column-major vector
    loads (n a multiple of the vector) and element loads (odd n) WITHOUT the row-major twin, 128-column panels of a
    wide block (contiguous offsets and column lists), row lists and column lists -- against explicit centring."""
    import tabmat_amd.dense_matrix as dmod
    from tabmat_amd.ext import dense as xd
    from tabmat_amd.ext._types import DenseDev

    monkeypatch.setattr(dmod, "ROW_MAJOR_TWIN", False)
    rng = np.random.default_rng(n + k)
    X = _dense_cols(rng, n, k, order).astype(dtype)
    c = np.round(X.astype(np.float64).mean(axis=0), 2).astype(dtype)
    Xc = (X - c[None, :]).astype(dtype).astype(np.float64)      # (the kernel rounds x - c to the block's dtype)
    d = rng.random(n).astype(dtype)
    d64 = d.astype(np.float64)
    dev = DenseDev.from_host(X)
    assert bool(dev.order_f) == (order == "F")
    dd, cd = torch.from_numpy(d).cuda(), torch.from_numpy(c).cuda()
    tol = 1e-12 if dtype == np.float64 else 2e-5
    want = (Xc.T * d64) @ Xc
    assert nat_err(xd.dense_sandwich(dev, dd, None, None, center=cd).cpu().numpy(), want) < tol
    rows = np.sort(rng.choice(n, n // 7, replace=False)).astype(np.int32)
    cols = np.sort(rng.choice(k, max(2, (2 * k) // 3), replace=False)).astype(np.int32)
    rd, cld = torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()
    w_r = (Xc[rows].T * d64[rows]) @ Xc[rows]
    assert nat_err(xd.dense_sandwich(dev, dd, rd, None, center=cd).cpu().numpy(), w_r) < tol
    w_c = (Xc[:, cols].T * d64) @ Xc[:, cols]
    assert nat_err(xd.dense_sandwich(dev, dd, None, cld, center=cd).cpu().numpy(), w_c) < tol
    w_rc = (Xc[np.ix_(rows, cols)].T * d64[rows]) @ Xc[np.ix_(rows, cols)]
    assert nat_err(xd.dense_sandwich(dev, dd, rd, cld, center=cd).cpu().numpy(), w_rc) < tol
