import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = 10_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
dm, sm = X.matrices[0], X.matrices[1]
oh, _ = X._onehot_slab([2, 3, 4])
_lib.call("tm_profile_enable", 1)
for name, slab in (("sparse", sm._slab()), ("onehot", oh)):
    for mode in (0, 1, 2, 3, 4, 7):
        os.environ["TM_GATHER_DBG"] = str(mode)
        ts = []
        for _ in range(3):
            xs.csr_dense_sandwich_slab(slab, dm._dev(), d)
            ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
        print(f"{name:7s} dbg={mode} (1=no compute,2=no global loads,4=no LDS store): {min(ts):8.3f} ms")
