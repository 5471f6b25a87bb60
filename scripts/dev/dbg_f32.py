import os, sys, torch, numpy as np, scipy.sparse as sps
sys.path.insert(0, "/root/repo")
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext._types import SlabEnt, CsrDev, DenseDev
torch.set_printoptions(linewidth=200, precision=4)
n, m, k = 64, 16, 128
rng = np.random.default_rng(0)
dts = torch.float32
B = (torch.arange(n, device="cuda", dtype=torch.float64)[:, None] * 1000 + torch.arange(k, device="cuda", dtype=torch.float64)[None, :]).to(dts).contiguous()
for ne in (2, 8):
    cells = rng.choice(n * m, size=ne, replace=False)
    r, c = cells // m, cells % m
    v = np.ones(ne)
    A = sps.csr_matrix((v, (r, c)), shape=(n, m)); A.sort_indices()
    csr = CsrDev(torch.tensor(A.data, dtype=dts, device="cuda"), torch.tensor(A.indices, dtype=torch.int32, device="cuda"),
                 torch.tensor(A.indptr, dtype=torch.int64, device="cuda"), n, m)
    ent = SlabEnt.from_csr(csr)
    dd = torch.ones(n, dtype=dts, device="cuda")
    out_k = torch.zeros((ent.mk, k), dtype=dts, device="cuda")
    from tabmat_amd._lib import call
    from tabmat_amd import _device as D
    call("tm_csr_dense_sandwich_ent_f32", D.p(ent.vals), D.p(ent.meta), D.p(ent.bstart), n, ent.mk, D.p(B), k, D.p(dd), D.p(out_k), D.p(None), D.stream_ptr())
    print("entries (row, col):", sorted(zip(r.tolist(), c.tolist())))
    print("meta:", [(x >> 4, x & 15) for x in ent.meta[:16].tolist()], "vals", ent.vals[:16].tolist())
    print("inv", ent.inv.tolist())
    for q in range(ent.mk):
        if out_k[q].abs().sum() != 0:
            print(f"  kernel row {q}: cols 0..5 = {out_k[q][:6].tolist()}  cols 64..66 = {out_k[q][64:67].tolist()}")
