"""Ingest cost at cfg4: time, HBM peak and resident bytes of every twin `to_device()` builds, after a
small warm-up matrix has loaded the torch sort / scan modules.  One line per stage."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tabmat_amd import synth  # noqa: E402
from tabmat_amd.ext import split as xsplit  # noqa: E402

N = int(os.environ.get("N", 10_000_000))
warm = synth.mixed_split(50_000, 128, 512, (256, 96, 32), 0.05, torch.float64, 1)
warm.to_device()
warm.sandwich(torch.rand(50_000, dtype=torch.float64, device="cuda"))
del warm
torch.cuda.synchronize()
torch.cuda.empty_cache()
X = synth.mixed_split(N, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
print(f"data resident: {base / 1e9:.2f} GB")
sm = X.matrices[1]


def stage(name, fn):
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    m0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"{name:28s} {dt:9.1f} ms   +{(torch.cuda.memory_allocated() - m0) / 1e9:6.2f} GB resident   "
          f"peak {torch.cuda.max_memory_allocated() / 1e9:6.2f} GB", flush=True)
    return dt


tot = 0.0
tot += stage("chunk_major", lambda: sm._dev().chunk_major())
tot += stage("chunk_col8", lambda: sm._dev().chunk_col8())
tot += stage("pair_blocks", lambda: sm._dev().pair_blocks())
tot += stage("entry twin", lambda: sm._ent())
tot += stage("rest of to_device()", lambda: X.to_device())
d = torch.rand(N, dtype=torch.float64, device="cuda")
tot += stage("first sandwich", lambda: X.sandwich(d))
stage("second sandwich", lambda: X.sandwich(d))
print(f"ingest total {tot:.1f} ms; resident {torch.cuda.memory_allocated() / 1e9:.2f} GB "
      f"(data {base / 1e9:.2f})")
v = torch.rand(X.shape[1], dtype=torch.float64, device="cuda")
stage("first matvec", lambda: X.matvec(v))
stage("first transpose_matvec", lambda: X.transpose_matvec(d))
print(f"resident after all three products {torch.cuda.memory_allocated() / 1e9:.2f} GB")
