"""Dense C-order matvec / transpose_matvec bandwidth by row length (same-box A/B:
TABMAT_AMD_LIB=<old .so> TABMAT_AMD_LIB_LAX=1)."""
import sys
import numpy as np
import torch
import tabmat_amd as tm
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext._types import DenseDev

dev = torch.device("cuda:0")
import os
if os.environ.get("RMV_FLAT_MAX"):
    from tabmat_amd._lib import call
    call("tm_tune_set", b"rmv_flat_max", int(os.environ["RMV_FLAT_MAX"]))
widths = [int(a) for a in sys.argv[1:]] or [3, 5, 8, 10, 12, 16, 20, 24, 31, 32, 40, 48, 63, 64, 100, 127, 128, 200, 500]
for dt in ((torch.float64,) if os.environ.get('MV_F64_ONLY') else (torch.float64, torch.float32)):
    for m in widths:
        n = int(float(os.environ.get('MV_BYTES', 6.4e8)) // (m * (8 if dt == torch.float64 else 4)))
        Xt = torch.randn(n, m, dtype=dt, device=dev)
        X = DenseDev.from_tensor(Xt)
        v = torch.randn(m, dtype=dt, device=dev)
        w = torch.randn(n, dtype=dt, device=dev)
        o1 = torch.zeros(n, dtype=dt, device=dev)
        o2 = torch.zeros(m, dtype=dt, device=dev)
        res = []
        for fn in (lambda: xd.dense_matvec(X, v, None, None, o1), lambda: xd.dense_rmatvec(X, w, None, None, o2)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            res.append((ms, Xt.numel() * Xt.element_size() / ms / 1e6))
        print(f"{str(dt)[6:]:8s} m={m:5d} n={n:9d}  matvec {res[0][0]:.4f} ms {res[0][1]:7.0f} GB/s   "
              f"rmatvec {res[1][0]:.4f} ms {res[1][1]:7.0f} GB/s", flush=True)
        del X, Xt, w, o1
