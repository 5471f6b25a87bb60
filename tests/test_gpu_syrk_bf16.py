"""-m gpu parity of K1d, the f32 dense syrk on the bf16 matrix cores (csrc/syrk_bf16.hip; reference:
ext/dense_helpers-tmpl.cpp:266-311): against the f64 oracle on the same f32 data at the tolerance of
the f32-MFMA path (2e-5 of max|out|: f32 products accumulated over <= 5e4 rows), its error next to
that path's, negative / zero weights, chunk (32 rows) and item (1024 rows) edges."""
import numpy as np
import pytest
import torch

from _gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import oracle as orc

    return orc


def _tune(key, value):
    from tabmat_amd import _lib

    _lib.call("tm_tune_set", key.encode(), int(value))


@pytest.fixture
def knobs():
    yield _tune
    for k, v in (("syrk_bf16", 1), ("bx_grid", 512)):
        _tune(k, v)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 1023, 1024, 1025, 4097, 50_000])
@pytest.mark.parametrize("m", [4, 132, 200, 256])
def test_bf16x3_vs_oracle(n, m):
    from tabmat_amd.ext import dense as xd
    from tabmat_amd.ext._types import DenseDev

    rng = np.random.default_rng(n * 7 + m)
    X = rng.standard_normal((n, m)).astype(np.float32)
    d = (rng.random(n) - 0.2).astype(np.float32)          # some negative weights
    d[::5] = 0.0
    out = xd.dense_sandwich_bf16x3(DenseDev.from_host(X), torch.from_numpy(d).cuda()).cpu().numpy()
    ref = _orc().dense_sandwich(X.astype(np.float64), d.astype(np.float64), None, None)
    assert out.dtype == np.float32
    assert rel_err(out, ref) < 2e-5
    assert np.array_equal(out, out.T)


def test_bf16x3_error_next_to_the_f32_mfma_path(knobs):
    """Same data through both kernels, errors against the f64 oracle relative to each entry's natural
    scale sqrt(S_ii S_jj), columns of mixed magnitude.  Both are f32 products accumulated in f32
    (measured: 6.6e-7 for the split, 8e-8 for the f32-input MFMA, whose K = 4 products are summed
    before rounding): the split must stay 10x below the 2e-5 bar of the f32 path."""
    import tabmat_amd as tm

    rng = np.random.default_rng(0)
    n, m = 200_000, 256
    X = (rng.standard_normal((n, m)) * rng.lognormal(0, 2, m)).astype(np.float32)   # columns of mixed scale
    d = rng.random(n).astype(np.float32)
    ref = _orc().dense_sandwich(X.astype(np.float64), d.astype(np.float64), None, None)
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))                          # entry-wise natural scale
    a = tm.DenseMatrix(X).sandwich(d)
    knobs("syrk_bf16", 0)
    b = tm.DenseMatrix(X).sandwich(d)
    ea = float((np.abs(a - ref) / scale).max())
    eb = float((np.abs(b - ref) / scale).max())
    print(f"bf16x3 {ea:.2e}  f32 mfma {eb:.2e}")
    assert ea < 2e-6 and eb < 2e-6


def test_bf16x3_is_the_default_for_wide_f32_blocks(knobs):
    import tabmat_amd as tm
    from tabmat_amd import _lib
    import ctypes as C

    rng = np.random.default_rng(1)
    X = rng.standard_normal((30_000, 256)).astype(np.float32)
    d = rng.random(30_000).astype(np.float32)
    _lib.call("tm_profile_enable", 1)
    try:
        res = tm.DenseMatrix(X).sandwich(d)
    finally:
        _lib.call("tm_profile_enable", 0)
    ref = _orc().dense_sandwich(X.astype(np.float64), d.astype(np.float64), None, None)
    assert rel_err(res, ref) < 2e-5
    # few workgroups, many items each
    knobs("bx_grid", 3)
    assert rel_err(tm.DenseMatrix(X).sandwich(d), ref) < 2e-5
