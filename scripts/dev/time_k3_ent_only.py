"""K3 entry-list kernel alone at cfg4 size (A/B of kernel builds through TABMAT_AMD_LIB).
usage: python scripts/dev/time_k3_ent_only.py [rows] [density] [csum]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext._types import SlabEnt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
dt = torch.float64
dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, dens, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
ent = SlabEnt.from_csr(sm._dev())
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(5):
    out = xs.csr_dense_sandwich_ent(ent, Bd, d)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
print(f"{os.path.basename(os.environ.get('TABMAT_AMD_LIB', 'default')):36s} ent: {min(ts):.3f} ms  (all: {' '.join(f'{t:.2f}' for t in ts)})", flush=True)
