"""CSR matvec / transpose_matvec at cfg4's sparse block: 16-bit column twin against int32 columns, interleaved rounds in one process."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
A = sm._dev()
v = torch.rand(512, dtype=torch.float64, device="cuda")
w = torch.rand(n, dtype=torch.float64, device="cuda")
_lib.call("tm_profile_enable", 1)
def t(f, k=12):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
o1 = torch.zeros(n, dtype=torch.float64, device="cuda"); o2 = torch.zeros(512, dtype=torch.float64, device="cuda")
for r in range(4):
    row = []
    for flag in (False, True):
        xs.CSR_U16 = flag
        row.append((t(lambda: xs.csr_matvec(A, v, None, None, o1)), t(lambda: xs.csc_rmatvec(A, w, None, None, o2))))
    print(f"round {r}: int32 matvec {row[0][0]:.3f} rmatvec {row[0][1]:.3f}   uint16 matvec {row[1][0]:.3f} rmatvec {row[1][1]:.3f}", flush=True)
