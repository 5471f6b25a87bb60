"""Twin build times in steady state (second block of the same size in one process: no first-use
module loads of the torch kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
n = 10_000_000
def t(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) * 1e3:9.1f} ms", flush=True)
for rep in range(2):
    sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003 + rep)
    print(f"-- block {rep}")
    t("chunk-major twin (K2)", lambda: sm._dev().chunk_major())
    t("lane-group twin (K3)", lambda: sm._lg())
    t("slab form (cat x sparse)", lambda: sm._slab())
    del sm
    torch.cuda.empty_cache()
