"""K2 (chunk-major kernel) with 8 / 4 / 2 slots per row and chunk over density x width.
usage: K2_SLOTS=s python scripts/dev/time_k2_slots.py"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
if os.environ.get("K2_SLOTS"):
    _lib.call("tm_tune_set", b"k2_slots", int(os.environ["K2_SLOTS"]))
n = 2_000_000
out = []
for m, dens in ((512, 0.10), (512, 0.075), (512, 0.05), (512, 0.025), (512, 0.015), (512, 0.01), (512, 0.005), (2048, 0.0125), (2048, 0.004), (4096, 0.002)):
    sm = synth.sparse_block(n, m, dens, torch.float64, 7)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    A = sm._dev()
    A.chunk_major()
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(4):
        xs.sparse_sandwich_chunked(A, d)
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    out.append(f"{m}@{dens * 100:g}%({m / 128 * 0 + 128 * dens:.2f}/chunk): {min(ts):7.3f}")
    del sm, A
    torch.cuda.empty_cache()
print(f"slots {os.environ.get('K2_SLOTS', 'auto'):>4s} | " + " | ".join(out), flush=True)
