"""Narrow dense sandwich (1..11 columns) by width, staged (LDS) against direct loads (tune knob syrk_narrow_staged)."""
import torch
from tabmat_amd._lib import call
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext._types import DenseDev

dev = torch.device("cuda:0")
for dt in (torch.float64, torch.float32):
    for m in range(3, 12):
        n = int(3.2e8 // (m * (8 if dt == torch.float64 else 4)))
        Xt = torch.randn(n, m, dtype=dt, device=dev)
        X = DenseDev.from_tensor(Xt)
        d = torch.rand(n, dtype=dt, device=dev)
        res = []
        for staged in (1, 0):
            call("tm_tune_set", b"syrk_narrow_staged", staged)
            for _ in range(3):
                xd.dense_sandwich(X, d, None, None)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                xd.dense_sandwich(X, d, None, None)
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 20)
        gb = (Xt.numel() + n) * Xt.element_size() / 1e6
        print(f"{str(dt)[6:]:8s} m={m:3d} n={n:9d}  staged {res[0]:.4f} ms {gb / res[0]:6.0f} GB/s   direct {res[1]:.4f} ms {gb / res[1]:6.0f} GB/s", flush=True)
