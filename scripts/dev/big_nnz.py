"""A sparse block with MORE than 2^31 nonzeros on one MI355X: SplitMatrix works on it in row parts
(tabmat_amd/split_matrix.py::_parts).  Checks: symmetry, trace and grand sum of the sparse self
sandwich against direct reductions over the CSR arrays, the cross term with the dense block against
a transpose_matvec identity.  usage: python scripts/dev/big_nnz.py [rows]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tabmat_amd import synth
from tabmat_amd.split_matrix import SplitMatrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 90_000_000
t0 = time.perf_counter()
sp = synth.sparse_block(n, 512, 0.05, torch.float64, 11)
de = synth.dense_block(n, 16, torch.float64, 12)
ca = synth.cat_block(n, 50, 13)
X = SplitMatrix([de, sp, ca])
A = sp._dev()
nnz = int(A.data.numel())
torch.cuda.synchronize()
print(f"built {n} rows, nnz {nnz} = 2^31 x {nnz / 2**31:.3f} in {time.perf_counter() - t0:.1f} s", flush=True)
d = torch.rand(n, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
H = X.sandwich(d)
torch.cuda.synchronize()
print(f"first sandwich (parts + twins) {time.perf_counter() - t0:.1f} s; parts: {[(a, b) for a, b, _ in X._parts()]}", flush=True)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); H = X.sandwich(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"sandwich {min(ts) * 1e3:.1f} ms  ({synth.algorithmic_bytes(X) / min(ts) / 1e9:.0f} GB/s effective)", flush=True)
S = H[16:16 + 512, 16:16 + 512]
counts = A.indptr[1:] - A.indptr[:-1]
rowid = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
dv = d[rowid]
trace_ref = (dv * A.data * A.data).sum().item()
rowsum = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, rowid, A.data)
grand_ref = (d * rowsum * rowsum).sum().item()
del rowid, dv
print(f"symmetric: {bool(torch.equal(H, H.T))}   trace rel.err {abs(S.trace().item() - trace_ref) / trace_ref:.2e}   "
      f"grand sum rel.err {abs(S.sum().item() - grand_ref) / grand_ref:.2e}", flush=True)
# cross term dense x sparse: 1' (De' D Sp) = (Sp' (d * De 1))'
w = d * de._dev().as_2d().sum(dim=1)
ref = sp.transpose_matvec(w)
got = H[:16, 16:16 + 512].sum(dim=0)
print(f"dense x sparse column sums rel.err {((got - ref).abs().max() / ref.abs().max()).item():.2e}", flush=True)
print(f"HBM in use {torch.cuda.memory_allocated() / 2**30:.0f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB")
