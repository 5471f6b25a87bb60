"""-m gpu parity tests, part 3: BASELINE.json's FULL sizes, through size-independent properties
(the oracle cannot run 10M rows in seconds): exact categorical counts, linearity in d, a
checksum of checksums that ties sandwich to matvec, symmetry, and a true oracle comparison on a
random row subset (rows=...) whose data is pulled back from HBM."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs
from _gpu_util import nat_err, rel_err

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc

    return orc


def _subset_specs(X, rows_t):
    """Pull the selected rows of a device SplitMatrix back as oracle blocks."""
    import tabmat_amd as tm

    blocks = []
    for m in X.matrices:
        if isinstance(m, tm.DenseMatrix):
            blocks.append(("dense", m._dev().as_2d()[rows_t].cpu().numpy()))
        elif isinstance(m, tm.SparseMatrix):
            c = m._dev()
            lo, hi = c.indptr[rows_t], c.indptr[rows_t + 1]
            cnt = (hi - lo)
            starts = torch.repeat_interleave(lo, cnt)
            offs = torch.arange(int(cnt.sum()), device=lo.device) - torch.repeat_interleave(
                torch.cumsum(cnt, 0) - cnt, cnt)
            sel = starts + offs
            indptr = np.concatenate([[0], np.cumsum(cnt.cpu().numpy())])
            S = sps.csr_matrix((c.data[sel].cpu().numpy(), c.indices[sel].cpu().numpy(), indptr),
                               shape=(len(rows_t), c.m))
            blocks.append(("sparse", S.tocsc()))
        else:
            blocks.append(("cat", m._dev()[rows_t].cpu().numpy(), m.shape[1] + int(m.drop_first),
                           m.drop_first))
    return blocks


def test_cfg4_split_sandwich_10M():
    """BASELINE configs[3]: dense 128 + sparse 512 @5% + cats (256, 96, 32), 10M rows, float64."""
    import tabmat_amd as tm
    from tabmat_amd import synth

    n = 10_000_000
    X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
    p = X.shape[1]
    assert p == 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    d1 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    d2 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    S1 = X.sandwich(d1)
    S2 = X.sandwich(d2)
    S12 = X.sandwich(d1 + d2)
    assert S1.dtype == torch.float64 and tuple(S1.shape) == (p, p)
    # symmetry: exact (mirrored tiles)
    assert torch.equal(S1, S1.T)
    # linearity in d
    assert float((S12 - (S1 + S2)).abs().max() / S12.abs().max()) < 1e-12
    # checksum of checksums: 1' (X' D X) 1 == sum_k d_k (X 1)_k^2, X 1 from the matvec kernels
    ones = torch.ones(p, dtype=torch.float64, device="cuda")
    rs = X.matvec(ones)
    lhs = float(S1.sum())
    rhs = float((d1 * rs * rs).sum())
    assert abs(lhs - rhs) / abs(rhs) < 1e-11
    # vector-level identity for random u: S u == X' (d * (X u)), the right side from the matvec /
    # transpose_matvec kernels (independent of every sandwich kernel).  Unlike the scalar checksum
    # this sees every entry of S with a random weight: an error confined to one tile of one block
    # product of the UNRESTRICTED kernels (the ones bench.py times) shows up in the rows of that tile.
    for seed in (11, 12, 13):
        gu = torch.Generator(device="cuda").manual_seed(seed)
        u = torch.randn(p, dtype=torch.float64, device="cuda", generator=gu)
        lhs_v = S1 @ u
        rhs_v = X.transpose_matvec(d1 * X.matvec(u))
        assert float((lhs_v - rhs_v).abs().max() / lhs_v.abs().max()) < 1e-11
    # X' d from transpose_matvec == first-moment identity through a second sandwich-free path
    tmv = X.transpose_matvec(d1)
    assert abs(float(tmv.sum()) - float((d1 * rs).sum())) / abs(float(tmv.sum())) < 1e-11
    # categorical diagonal blocks with d == 1 are exact counts
    Sones = X.sandwich(torch.ones(n, dtype=torch.float64, device="cuda"))
    for m, idx in zip(X.matrices, X.indices):
        if isinstance(m, tm.CategoricalMatrix):
            counts = torch.bincount(m._dev().to(torch.int64), minlength=m.shape[1]).to(torch.float64)
            ii = torch.as_tensor(idx, device="cuda")
            assert torch.equal(Sones[ii, ii], counts)
    # column selections at full size: a narrow one (dense-block form of the selected columns), a
    # wide one (unrestricted product + selection) and the identity must all be sub-blocks of S1
    rng = np.random.default_rng(0)
    for share in (0.04, 0.7):
        cols = np.sort(rng.choice(p, size=int(share * p), replace=False)).astype(np.int32)
        ct = torch.as_tensor(cols.astype(np.int64), device="cuda")
        Sc = X.sandwich(d1, cols=cols)
        want = S1[ct][:, ct]
        assert float((Sc - want).abs().max() / want.abs().max()) < 1e-12
        assert float((X.transpose_matvec(d1, cols=cols) - tmv[ct]).abs().max() / tmv.abs().max()) < 1e-12
        vm = torch.zeros(p, dtype=torch.float64, device="cuda")
        vm[ct] = 1.0
        assert float((X.matvec(ones, cols=cols) - X.matvec(vm)).abs().max()) < 1e-9
    assert torch.equal(X.sandwich(d1, cols=np.arange(p)), S1) or \
        float((X.sandwich(d1, cols=np.arange(p)) - S1).abs().max() / S1.abs().max()) < 1e-12
    # true oracle comparison of the UNRESTRICTED path on contiguous device slices X[lo:hi] of 50k
    # rows, taken at three offsets >= 2^31 bytes into the dense block (row 2^21 and beyond): the
    # slices stay in HBM (device __getitem__), run the unrestricted kernels, and are compared with
    # the oracle on the pulled-back rows -- one by one and summed.
    tot_gpu, tot_ref = None, None
    for lo in (2_200_000, 5_000_017, 9_949_999):
        hi = lo + 50_000
        assert lo * 128 * 8 >= 2 ** 31
        Xs = X[lo:hi]
        ds = d1[lo:hi].contiguous()
        got = Xs.sandwich(ds).cpu().numpy()
        rows_t = torch.arange(lo, hi, device="cuda")
        blocks = [cs.to_oracle_block(sp) for sp in _subset_specs(X, rows_t)]
        ref = _orc().split_sandwich(blocks, [np.asarray(i) for i in X.indices], ds.cpu().numpy())
        assert rel_err(got, ref) < 1e-10
        assert nat_err(got, ref) < 1e-10          # entry by entry at the natural scale sqrt(S_ii S_jj)
        tot_gpu = got if tot_gpu is None else tot_gpu + got
        tot_ref = ref if tot_ref is None else tot_ref + ref
    assert rel_err(tot_gpu, tot_ref) < 1e-10
    assert nat_err(tot_gpu, tot_ref) < 1e-10
    # true oracle comparison on a random row subset via rows=
    rows = np.sort(rng.choice(n, size=20_000, replace=False)).astype(np.int32)
    rows_t = torch.as_tensor(rows.astype(np.int64), device="cuda")
    sub = X.sandwich(d1, rows=rows)
    blocks = [cs.to_oracle_block(s) for s in _subset_specs(X, rows_t)]
    ref = _orc().split_sandwich(blocks, [np.asarray(i) for i in X.indices], d1[rows_t].cpu().numpy())
    assert rel_err(sub.cpu().numpy(), ref) < 1e-10
    assert nat_err(sub.cpu().numpy(), ref) < 1e-10


@pytest.mark.parametrize("order", ["C", "F"])
def test_cfg1_dense_f64_100k_x_64(order):
    """BASELINE configs[0] at its exact workload (X = default_rng(0).standard_normal((100_000, 64)),
    float64, d = rng.random(n)): DenseMatrix.sandwich against the oracle's restatement of the reference loop
    (dense_helpers-tmpl.cpp:198-417; the reference's own test of this path is tests/test_fast_sandwich.py:51-98) at
    FULL size -- the oracle needs milliseconds for it.  (64 columns are below the int8-sliced syrk's 65-column floor:
    this block runs on the float64 MFMA syrk whatever set_strict_f64 says.)  Both memory orders, with and without
    rows / cols; also the two matrix-vector products of the block."""
    import tabmat_amd as tm

    rng = np.random.default_rng(0)
    n, k = 100_000, 64
    X = rng.standard_normal((n, k))
    d = rng.random(n)
    Xo = np.asfortranarray(X) if order == "F" else X
    dm = tm.DenseMatrix(Xo)
    orc = _orc()
    want = orc.dense_sandwich(X, d, None, None)
    got = dm.sandwich(d)
    assert got.dtype == np.float64 and got.shape == (k, k)
    assert nat_err(got, want) < 1e-10 and rel_err(got, want) < 1e-10
    rows = np.sort(rng.choice(n, 40_000, replace=False)).astype(np.int32)
    cols = np.sort(rng.choice(k, 41, replace=False)).astype(np.int32)
    for r, c in ((rows, None), (None, cols), (rows[:900], cols)):
        w = orc.dense_sandwich(X, d, r, c)
        assert nat_err(dm.sandwich(d, rows=r, cols=c), w) < 1e-10
    v = rng.standard_normal(k)
    assert rel_err(dm.matvec(v), X @ v) < 1e-12
    assert rel_err(dm.transpose_matvec(d), X.T @ d) < 1e-12


def test_cfg2_dense_f32_10M_x_256():
    """BASELINE configs[1]: DenseMatrix.sandwich float32, 10M x 256 (MFMA row-weighted syrk)."""
    from tabmat_amd import synth

    n, k = 10_000_000, 256
    X = synth.dense_block(n, k, torch.float32, seed=1)
    g = torch.Generator(device="cuda").manual_seed(2)
    d1 = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
    d2 = torch.rand(n, dtype=torch.float32, device="cuda", generator=g)
    S1, S2, S12 = X.sandwich(d1), X.sandwich(d2), X.sandwich(d1 + d2)
    assert S1.dtype == torch.float32 and tuple(S1.shape) == (k, k)
    assert torch.equal(S1, S1.T)
    scale = float(S12.abs().max())
    # fp32 accumulation over ~2e4 rows per workgroup, partials combined in double: 1e-4 relative
    assert float((S12 - (S1 + S2)).abs().max()) / scale < 1e-4
    # trace identity against an independent fp64 torch reduction
    A = X._dev().as_2d()
    tr = 0.0
    for lo in range(0, n, 2_000_000):
        blk = A[lo:lo + 2_000_000].to(torch.float64)
        tr += float(((blk * blk).sum(dim=1) * d1[lo:lo + 2_000_000].to(torch.float64)).sum())
    assert abs(float(S1.to(torch.float64).diagonal().sum()) - tr) / tr < 1e-5
    # oracle on a row subset
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(n, size=30_000, replace=False)).astype(np.int32)
    rt = torch.as_tensor(rows.astype(np.int64), device="cuda")
    ref = _orc().dense_sandwich(A[rt].cpu().numpy().astype(np.float64),
                                d1[rt].cpu().numpy().astype(np.float64), None, None)
    assert rel_err(X.sandwich(d1, rows=rows).cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize("zipf", [0.0, 1.1])
def test_cfg3_categorical_50M_x_10k(zipf):
    """BASELINE configs[2]: CategoricalMatrix.sandwich + transpose_matvec, 50M rows x 10k
    categories (uniform and Zipf-skewed codes).  Counts are bit-exact."""
    from tabmat_amd import synth

    n, c = 50_000_000, 10_000
    X = synth.cat_block(n, c, seed=2, zipf=zipf)
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    counts = torch.bincount(X._dev().to(torch.int64), minlength=c).to(torch.float64)
    diag = X._sandwich_diag_dev(ones, None, None)
    assert torch.equal(diag, counts)
    assert torch.equal(X.transpose_matvec(ones), counts)
    g = torch.Generator(device="cuda").manual_seed(3)
    d = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    dd = X._sandwich_diag_dev(d, None, None)
    # independent reference on the host (torch.index_add_ degenerates under skewed indices)
    ref = np.bincount(X._dev().cpu().numpy(), weights=d.cpu().numpy(), minlength=c)
    assert rel_err(dd.cpu().numpy(), ref) < 1e-12
    assert abs(float(dd.sum()) - float(d.sum())) / float(d.sum()) < 1e-12
    # matvec (gather) is exact
    v = torch.rand(c, dtype=torch.float64, device="cuda", generator=g)
    assert torch.equal(X.matvec(v), v[X._dev().to(torch.int64)])
