"""Time of the one-off device work before the first sandwich at cfg4 (10M rows): the twins of the
sparse block (chunk-major, lane-group, slab), the row-major dense view, the first call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
n = 10_000_000
X = synth.mixed_split(n)
d = torch.rand(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
sm = X.matrices[1]
def t(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) * 1e3:9.1f} ms", flush=True)
t("chunk-major twin (K2)", lambda: sm._dev().chunk_major())
t("lane-group twin (K3)", lambda: sm._lg())
t("slab form (cat x sparse)", lambda: sm._slab())
t("first sandwich (rest of ingest)", lambda: X.sandwich(d))
t("second sandwich", lambda: X.sandwich(d))
print(f"max memory allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, now {torch.cuda.memory_allocated() / 2**30:.1f} GiB")
