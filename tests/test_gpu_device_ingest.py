"""from_csc on storage that already lives in HBM (reference constructor.py:297-308,
constructor_util.py:11-49): the device split must equal the host split block by block, and the
products of the resulting SplitMatrix must equal the oracle's."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

import _cases as cs
from _gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _mixed_csc(n, m, seed):
    rng = np.random.default_rng(seed)
    dens = rng.choice([0.01, 0.05, 0.3, 0.9], size=m)
    A = np.where(rng.random((n, m)) < dens, rng.standard_normal((n, m)), 0.0)
    return sps.csc_matrix(A), A


@pytest.mark.parametrize("how", ["device_sparse_matrix", "raw_device_arrays"])
@pytest.mark.parametrize("threshold", [0.1, 0.5, 0.0, 1.0])
def test_device_split_equals_host_split(how, threshold):
    import tabmat_amd as tm
    from tabmat_amd.constructor import csc_arrays_to_csr_dev, from_csc

    S, A = _mixed_csc(3001, 37, seed=7)
    host = from_csc(S, threshold)
    if how == "raw_device_arrays":
        dev = from_csc((torch.from_numpy(S.data).cuda(), torch.from_numpy(S.indices).cuda(),
                        torch.from_numpy(S.indptr).cuda(), S.shape), threshold)
    else:
        csr = csc_arrays_to_csr_dev(S.data, S.indices, S.indptr, S.shape)
        sm = tm.SparseMatrix.from_device(csr)
        assert sm._array is None                      # nothing on the host
        dev = from_csc(sm, threshold)
        assert sm._array is None                      # ... and the split did not pull it back
    assert [type(b).__name__ for b in dev.matrices] == [type(b).__name__ for b in host.matrices]
    for a, b in zip(dev.indices, host.indices):
        assert np.array_equal(a, b)
    for a, b in zip(dev.matrices, host.matrices):
        assert a.shape == b.shape
        assert np.array_equal(np.asarray(a.toarray()), np.asarray(b.toarray()))
    rng = np.random.default_rng(1)
    d = rng.random(S.shape[0])
    v = rng.standard_normal(S.shape[1])
    assert rel_err(dev.sandwich(d), (A.T * d) @ A) < 1e-10
    assert rel_err(dev.matvec(v), A @ v) < 1e-10
    assert rel_err(dev.transpose_matvec(d), A.T @ d) < 1e-10


def test_device_split_keeps_names():
    import tabmat_amd as tm
    from tabmat_amd.constructor import csc_arrays_to_csr_dev, from_csc

    S, _ = _mixed_csc(500, 9, seed=3)
    names = [f"c{i}" for i in range(9)]
    host = from_csc(S, 0.1, column_names=names)
    csr = csc_arrays_to_csr_dev(S.data, S.indices, S.indptr, S.shape)
    dev = from_csc(tm.SparseMatrix.from_device(csr), 0.1, column_names=names)
    assert dev.get_names("column") == host.get_names("column")


@pytest.mark.parametrize("idx_dtype,ptr_dtype", [("int64", "int64"), ("int64", "int32"), ("int32", "int32")])
def test_device_csr_arrays_of_either_index_width(idx_dtype, ptr_dtype):
    """The reference's sparse kernels take int32 or int64 indices (`win_integral`, ext/sparse.pyx:13-15).  Device CSR
    arrays of either width are taken without a host round trip: tm_index_narrow_i64 / tm_index_widen_i32."""
    import torch
    from scipy import sparse as sps

    import tabmat_amd as tm
    from tabmat_amd.ext._types import CsrDev

    rng = np.random.default_rng(11)
    S = sps.random(4_000, 300, density=0.05, format="csr", random_state=rng, dtype=np.float64)
    S.sort_indices()
    data = torch.from_numpy(S.data).cuda()
    ind = torch.from_numpy(S.indices.astype(idx_dtype)).cuda()
    ptr = torch.from_numpy(S.indptr.astype(ptr_dtype)).cuda()
    sm = tm.SparseMatrix.from_device(CsrDev.from_device_arrays(data, ind, ptr, S.shape))
    d = rng.random(4_000)
    v = rng.random(300)
    assert np.allclose(sm.matvec(v), S @ v, rtol=1e-12, atol=1e-12)
    assert np.allclose(sm.transpose_matvec(d), S.T @ d, rtol=1e-12, atol=1e-12)
    assert np.allclose(sm.sandwich(d), (S.T.multiply(d) @ S).toarray(), rtol=1e-10, atol=1e-12)
    if idx_dtype == "int64":
        bad = ind.clone()
        bad[17] = 300                      # == m: outside [0, m)
        with pytest.raises(ValueError):
            CsrDev.from_device_arrays(data, bad, ptr, S.shape)
        bad[17] = -1
        with pytest.raises(ValueError):
            CsrDev.from_device_arrays(data, bad, ptr, S.shape)
