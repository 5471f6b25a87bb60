"""-m gpu parity tests of K1c, the LDS-light dense syrk with dynamic work items (csrc/syrk_co.hip;
reference: ext/dense_helpers-tmpl.cpp:266-311): against the CPU oracle at every edge of its chunk /
item / column-block geometry, its X'd side output (standardized_mat.py:149-150), the guest-stream
form of SplitMatrix.sandwich, and the workgroup placement log."""
import numpy as np
import pytest

import _cases as cs
from _gpu_util import rel_err, to_tm_split

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10


def _orc():
    from oracle import oracle as orc

    return orc


def _tune(key, value):
    from tabmat_amd import _lib

    _lib.call("tm_tune_set", key.encode(), int(value))


@pytest.fixture
def knobs(monkeypatch):
    # (the int8-sliced syrk would take these blocks first: K1c is its fallback and the X'd form)
    import tabmat_amd.dense_matrix as dmod

    monkeypatch.setattr(dmod, "SYRK_I8", False)
    yield _tune
    for k, v in (("co_grid", 768), ("syrk_co", 1), ("wg_log", 0)):
        _tune(k, v)


# chunk = 12 rows, item = 768 rows: sizes around both, column counts around the 16 / 32-column
# virtual blocks; odd counts since round 5 (the pair that straddles a row's end; the single-element load behind
# the last row)
@pytest.mark.parametrize("n", [1, 11, 12, 13, 767, 768, 769, 1537, 5000, 100003])
@pytest.mark.parametrize("m", [2, 16, 30, 34, 64, 100, 128, 1, 3, 33, 65, 101, 127])
def test_syrk_co_vs_oracle(n, m):
    import torch

    from tabmat_amd.ext import dense as xd
    from tabmat_amd.ext._types import DenseDev

    rng = np.random.default_rng(n * 131 + m)
    X = rng.standard_normal((n, m))
    d = rng.random(n)
    Xd = DenseDev.from_host(X)
    dd = torch.from_numpy(d).cuda()
    assert xd.co_supported(Xd, dd, any_width=True)
    out, csum = xd.dense_sandwich_co(Xd, dd, want_colsum=True)
    out, csum = out.cpu().numpy(), csum.cpu().numpy()
    ref = _orc().dense_sandwich(X, d, None, None)
    assert rel_err(out, ref) < F64_TOL
    assert np.array_equal(out, out.T)
    assert rel_err(csum, X.T @ d) < F64_TOL


@pytest.mark.parametrize("grid", [1, 2, 7, 256])
def test_syrk_co_work_items(grid, knobs):
    """Few workgroups, many items each: every workgroup walks several items fetched from the
    atomic counter (item boundaries inside the double-buffered chunk stream)."""
    import tabmat_amd as tm

    knobs("co_grid", grid)
    rng = np.random.default_rng(grid)
    n, m = 20011, 96
    X = rng.standard_normal((n, m))
    d = rng.random(n) - 0.3                      # negative weights are legal
    res = tm.DenseMatrix(X).sandwich(d)
    assert rel_err(res, _orc().dense_sandwich(X, d, None, None)) < F64_TOL
    assert np.array_equal(res, res.T)


def test_syrk_co_is_the_default_f64_path_and_matches_plain(knobs):
    import tabmat_amd as tm

    rng = np.random.default_rng(3)
    X = rng.standard_normal((30011, 128))
    d = rng.random(30011)
    a = tm.DenseMatrix(X).sandwich(d)
    knobs("syrk_co", 0)
    b = tm.DenseMatrix(X).sandwich(d)
    ref = _orc().dense_sandwich(X, d, None, None)
    assert rel_err(a, ref) < F64_TOL and rel_err(b, ref) < F64_TOL
    assert rel_err(a, b) < 1e-13


def test_zero_weight_rows_and_empty(knobs):
    import tabmat_amd as tm

    rng = np.random.default_rng(4)
    X = rng.standard_normal((4000, 64))
    d = rng.random(4000)
    d[::3] = 0.0
    assert rel_err(tm.DenseMatrix(X).sandwich(d), (X.T * d) @ X) < F64_TOL
    assert tm.DenseMatrix(np.zeros((0, 64))).sandwich(np.zeros(0)).shape == (64, 64)


def test_placement_log(knobs):
    """The instrumented kernels append one record per workgroup to the "wg_log" buffer."""
    import torch

    import tabmat_amd as tm

    cap = 4096
    buf = torch.zeros((cap + 1, 4), dtype=torch.int64, device="cuda")
    buf[0, 1] = cap
    knobs("wg_log", buf.data_ptr())
    knobs("co_grid", 64)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((100_000, 128))
    tm.DenseMatrix(X).sandwich(rng.random(100_000))
    torch.cuda.synchronize()
    knobs("wg_log", 0)
    L = buf.cpu().numpy()
    count = int(L[0, 0])
    assert count == 64
    rec = L[1:1 + count]
    assert (rec[:, 3] == 1).all()                      # WG_SYRK_CO
    assert (rec[:, 2] >= rec[:, 1]).all()              # end >= start
