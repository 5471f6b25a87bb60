"""K3 (sparse x dense 128) at low densities: interleaved-ELL twin vs compact slab stream."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
import tabmat_amd.sparse_matrix as smod
n = 10_000_000
for dens in (float(x) for x in os.environ.get("DENS", "0.005,0.0005").split(",")):
    for pad in (1e9, 0.0):
        smod.ELL_MAX_PAD = pad
        X = synth.mixed_split(n, 128, 512, (8,), dens, torch.float64, 3)
        d = torch.rand(n, dtype=torch.float64, device="cuda")
        dm, sm = X.matrices[0], X.matrices[1]
        _lib.call("tm_profile_enable", 1)
        ts = []
        for _ in range(3):
            sm._cross_sandwich_dev(dm, d, None, None, None)
            ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
        ell = sm._ell(wide=True)
        print(f"K3 density {dens} {'ELL' if ell is not None else 'slab'}: {min(ts):.3f} ms"
              + (f"  (ELL slots / nnz = {ell.vals.numel() / max(1, sm._dev().data.numel()):.1f})" if ell is not None else ""))
        del X, sm, dm
        torch.cuda.empty_cache()
