"""Lane-group twin of a sparse block (tabmat_amd/ext/_types.py::SlabLg, csrc/sparse_lg.hip).

CPU part: the twin builder is plain torch, so its output is decoded here exactly the way the
kernel walks it (blocks, rounds, chunks, slots, koff -> slab row) and compared with dense
algebra.  GPU part: the kernel itself against the oracle's csr_dense_sandwich."""
import numpy as np
import pytest
import torch
from scipy import sparse as sps

from tabmat_amd.ext._types import CsrDev, SlabLg

HAS_GPU = torch.cuda.is_available()


def _csr_cpu(S, dtype):
    S = sps.csr_matrix(S).astype(dtype)
    S.sort_indices()
    return CsrDev(torch.from_numpy(S.data.copy()), torch.from_numpy(S.indices.astype(np.int32)),
                  torch.from_numpy(S.indptr.astype(np.int64)), S.shape[0], S.shape[1])


def _decode(tw: SlabLg, itemsize):
    """(kernel column, row, value) triplets as the kernel reads them."""
    R, C, CH, SL = 64, 16, 4, 32
    rowb = 1024 if itemsize == 8 else 512      # tm_lg_row_bytes: the LDS row stride
    G = tw.mk // C
    S = (tw.n + R - 1) // R
    vals, koff = tw.vals.numpy(), tw.koff.numpy().view(np.uint32)
    xvals, xkoff, xptr = tw.xvals.numpy(), tw.xkoff.numpy().view(np.uint32), tw.xptr.numpy()
    out = []
    for blk in range(S * G):
        s, g = divmod(blk, G)
        base = blk * CH * SL
        nrec = int(koff[base] >> 20)
        assert xptr[blk + 1] - xptr[blk] == nrec
        if nrec:
            rec0 = int(koff[base + 1] >> 20) | int(koff[base + 2] >> 20) << 12 | int(koff[base + 3] >> 20) << 24
            assert rec0 == xptr[blk]
        for c in range(CH):         # round 0: four chunks at the fixed stride
            for slot in range(SL):
                q = (blk * CH + c) * SL + slot
                k = int(koff[q]) & 0xFFFFF
                if k == 0:
                    assert vals[q] == 0
                    continue
                h, jl, it = slot // 16, (slot % 16) // 8, slot % 8
                assert k % rowb == 0
                out.append((g * C + 8 * h + 2 * c + jl, s * R + k // rowb - 1, float(vals[q]), it))
        seen = {}
        for t in range(nrec):       # overflow entries of the block: {value, koff, column}
            e = xkoff[(int(xptr[blk]) + t) * 4:(int(xptr[blk]) + t) * 4 + 4]
            val = e[:2].view(np.float64)[0] if itemsize == 8 else e[:1].view(np.float32)[0]
            k = int(e[2]) & 0xFFFFF
            assert k and k % rowb == 0
            w = int(e[3]) & 15
            pos = 8 + seen.get(w, 0)
            seen[w] = seen.get(w, 0) + 1
            out.append((g * C + w, s * R + k // rowb - 1, float(val), pos))
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,density", [(300, 40, 0.05), (64, 16, 0.5), (129, 33, 0.2),
                                         (1000, 512, 0.02), (5, 3, 1.0)])
def test_twin_decodes_to_the_matrix(n, m, density, dtype):
    rng = np.random.default_rng(n + m)
    S = sps.random(n, m, density=density, format="csr", random_state=rng, dtype=np.float64)
    S.data += 0.5          # no explicit zeros
    tw = SlabLg.from_csr(_csr_cpu(S, dtype), max_pad=None, max_extra=None)
    inv = tw.inv.numpy()
    dense = np.zeros((tw.mk, n))
    seen = {}
    for kc, row, v, pos in _decode(tw, np.dtype(dtype).itemsize):
        assert dense[kc, row] == 0
        dense[kc, row] = v
        seen.setdefault((kc, row // 64), []).append((pos, row))
    np.testing.assert_array_equal(dense[inv].T, S.toarray().astype(dtype))
    for lst in seen.values():      # positions compacted, rows ascending with the position
        lst.sort()
        assert [p for p, _ in lst] == list(range(len(lst)))
        assert [r for _, r in lst] == sorted(r for _, r in lst)
    assert tw.unc in (2, 4)


def _decode_compact(tw: SlabLg):
    """(kernel column, row, value, position) of round 0 as the kernel reads the COMPACT stream."""
    R, C, CH, SL = 64, 16, 4, 32
    G = tw.mk // C
    S = (tw.n + R - 1) // R
    cvals, cmap, crec = tw.cvals.numpy(), tw.cmap.numpy(), tw.crec.numpy()
    assert cmap.shape == (S * G, SL, CH) and crec.shape == (S * G, 2)
    out = []
    for blk in range(S * G):
        s, g = divmod(blk, G)
        nxt = int(crec[blk, 0])
        for c in range(CH):
            for slot in range(SL):
                b = int(cmap[blk, slot, c])
                if b == 0:
                    continue
                h, jl, it = slot // 16, (slot % 16) // 8, slot % 8
                out.append((g * C + 8 * h + 2 * c + jl, s * R + b - 1, float(cvals[nxt]), it))
                nxt += 1
        assert nxt == (int(crec[blk + 1, 0]) if blk + 1 < S * G else len(cvals) - 1)
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,density", [(300, 40, 0.05), (64, 16, 0.5), (129, 33, 0.2), (1000, 512, 0.02)])
def test_compact_twin_holds_the_same_round_0(n, m, density, dtype):
    """compact_(): values of the real slots only + a byte per slot + a record per block -- the same
    (column, row, value, position) set as the padded stream, the overflow entries untouched."""
    rng = np.random.default_rng(n + 3 * m)
    S = sps.random(n, m, density=density, format="csr", random_state=rng, dtype=np.float64)
    S.data += 0.5
    tw = SlabLg.from_csr(_csr_cpu(S, dtype), max_pad=None, max_extra=None)
    padded = [t for t in _decode(tw, np.dtype(dtype).itemsize) if t[3] < 8]
    xptr = tw.xptr.numpy().copy()
    tw.compact_()
    assert tw.vals is None and tw.koff is None and tw.dtype == torch.from_numpy(np.zeros(1, dtype)).dtype
    assert sorted(_decode_compact(tw)) == sorted(padded)
    rec = tw.crec.numpy()[:, 1]
    np.testing.assert_array_equal(rec & 0xFFFFFFFF, xptr[1:] - xptr[:-1])
    np.testing.assert_array_equal((rec >> 32)[(rec & 0xFFFFFFFF) > 0], xptr[:-1][(rec & 0xFFFFFFFF) > 0])


def test_twin_gives_up_outside_its_regime():
    rng = np.random.default_rng(0)
    S = sps.random(64 * 1200, 512, density=0.0005, format="csr", random_state=rng)
    assert SlabLg.from_csr(_csr_cpu(S, np.float64), max_pad=8.0) is None          # too sparse
    S = sps.random(64 * 1200, 512, density=0.3, format="csr", random_state=rng)
    assert SlabLg.from_csr(_csr_cpu(S, np.float64), max_pad=8.0) is None          # too dense
    assert SlabLg.from_csr(_csr_cpu(S, np.float64), max_pad=8.0, max_extra=None) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("unc", [2, 4])
@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("n,m,k,density", [(20_000, 512, 128, 0.05), (777, 40, 72, 0.1),
                                           (64, 16, 128, 0.5), (4099, 100, 200, 0.3),
                                           (30_000, 300, 256, 0.01), (129, 7, 68, 1.0)])
def test_lg_kernel_matches_oracle(n, m, k, density, unc, compact, dtype):
    import tabmat_amd as tm
    from oracle import oracle as orc
    from tabmat_amd import _device as D
    from tabmat_amd.ext import sparse as xs

    rng = np.random.default_rng(n + m + k)
    S = sps.random(n, m, density=density, format="csc", random_state=rng, dtype=np.float64)
    S.data -= 0.5
    S = S.astype(dtype)
    B = rng.standard_normal((n, k)).astype(dtype)
    d = rng.random(n).astype(dtype)
    d[rng.integers(0, n, n // 7)] = 0          # d == 0 rows take the zero-row redirect
    B[d == 0] = np.inf                           # ... and must not leak inf * 0
    sm, dm = tm.SparseMatrix(S), tm.DenseMatrix(B)
    tw = SlabLg.from_csr(sm._dev(), max_pad=None, max_extra=None)
    if compact:
        tw.compact_()
    got = D.to_host(xs.csr_dense_sandwich_lg(tw, dm._dev_c(), D.to_dev(d), unc=unc))
    Bz = B.copy()
    Bz[d == 0] = 0
    want = orc.csr_dense_sandwich(sps.csr_matrix(S), Bz, d, None, None, None)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    scale = max(np.abs(want).max(), 1e-30)
    assert np.abs(got - want).max() / scale < tol
