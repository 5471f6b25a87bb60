"""Time the K3 lane-group kernel at cfg4 size with the library named by TABMAT_AMD_LIB (kernel
experiments built into tabmat_amd/_abl/*.so).  usage: TABMAT_AMD_LIB=... python time_k3_variants.py"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = 10_000_000
dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, 0.05, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
lg = sm._lg()
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(6):
    out = xs.csr_dense_sandwich_lg(lg, Bd, d)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"{os.environ.get('TABMAT_AMD_LIB', 'default'):40s} min {min(ts):.3f} ms  median {sorted(ts)[3]:.3f}  checksum {out.double().sum().item():.6e}", flush=True)
