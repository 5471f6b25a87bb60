"""2M x 2048 @ 5 %: both sparse self-sandwich kernels against a dense product of a few column pairs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.ext import sparse as xs
n, m, dens = 2_000_000, 2048, 0.05
sm = synth.sparse_block(n, m, dens, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
A = sm._dev()
cols = torch.tensor([0, 1, 17, 127, 128, 500, 1023, 1024, 1500, 2047], device="cuda")
# dense copies of those columns
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), A.indptr[1:] - A.indptr[:-1])
T = torch.zeros((n, len(cols)), dtype=torch.float64, device="cuda")
for k, c in enumerate(cols.tolist()):
    sel = A.indices == c
    T[rows[sel], k] = A.data[sel]
want = T.T @ (T * d[:, None])
cur = sm._sandwich_dev(d, None, None)[cols][:, cols]
new = xs.sparse_sandwich_pairs(A, d)[cols][:, cols]
dg = torch.sqrt(torch.diagonal(want)); den = torch.outer(dg, dg)
print("current path err", float(((cur - want).abs() / den).max()), " pairs kernel err", float(((new - want).abs() / den).max()))
full = xs.sparse_sandwich_pairs(A, d)
dgf = torch.diagonal(full)
print("zero diagonal entries:", int((dgf == 0).sum()), "of", m, " first few:", torch.nonzero(dgf == 0).flatten()[:20].tolist())
a_rec, a_pos, sptr, b_rec, bptr, tiles, tile_of, n_tiles, SW, JW = A.strip_twin()
print("n_tiles", n_tiles, "SW", SW, "JW", JW, "sptr last", sptr[-1, -5:].tolist(), "nnz", A.data.numel())
print("sptr monotone:", bool((sptr.flatten()[1:] >= sptr.flatten()[:-1]).all()), " strip ends == next start:", bool((sptr[:-1, -1] == sptr[1:, 0]).all()))
st = torch.div(a_rec[:, 2], SW, rounding_mode="floor")
print("strip order monotone:", bool((st[1:] >= st[:-1]).all()))
cnt = torch.bincount(st.to(torch.int64), minlength=128)
cs = torch.cumsum(cnt, 0) - cnt
print("sptr[:,0] == strip starts:", bool((sptr[:, 0].to(torch.int64) == cs).all()), sptr[20:26, 0].tolist(), cs[20:26].tolist())
rw = a_rec[:, 3]
same = st[1:] == st[:-1]
print("rows monotone inside strips:", bool((rw[1:][same] >= rw[:-1][same]).all()))
print("a_pos consistent:", bool((b_rec[a_pos.to(torch.int64), 2] == a_rec[:, 2]).all()))
