"""CSR matvec / transpose_matvec stream kernels at cfg4's sparse block (kernel time by HIP events)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
v = torch.rand(512, dtype=torch.float64, device="cuda"); w = torch.rand(n, dtype=torch.float64, device="cuda")
_lib.call("tm_profile_enable", 1)
def t(f, k=25):
    ts = []
    for _ in range(k):
        f(); ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    return min(ts)
A = sm._dev(); by = A.data.numel() * 12 + (n + 1) * 8 + n * 8 + 512 * 8
a = t(lambda: sm._matvec_dev(v, None, None, None, False)); b = t(lambda: sm._matvec_dev(w, None, None, None, True))
print(f"{os.environ.get('TABMAT_AMD_LIB', 'default'):40s} matvec {a:.3f} ms {by / a / 1e6:.0f} GB/s {by / a / 8e9:.3f} of HBM peak   rmatvec {b:.3f} ms {by / b / 1e6:.0f} GB/s {by / b / 8e9:.3f} of HBM peak")
