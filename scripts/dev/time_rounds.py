"""cfg4 step with 1 / 2 rounds of workgroups in the fused categorical cross kernels (tm_tune_set catdense_rounds / catsparse_rounds)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = 10_000_000
mat = synth.mixed_split(n)
d = torch.rand(n, dtype=torch.float64, device="cuda")
def t():
    for _ in range(3): mat.sandwich(d)
    torch.cuda.synchronize(); ts = []
    for _ in range(8):
        t0 = time.perf_counter(); mat.sandwich(d); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[4]
for cd, csp in ((1, 1), (2, 1), (1, 2), (2, 2), (4, 4), (1, 1)):
    _lib.call("tm_tune_set", b"catdense_rounds", cd)
    _lib.call("tm_tune_set", b"catsparse_rounds", csp)
    a, b = t()
    print(f"catdense_rounds {cd} catsparse_rounds {csp}: step min {a:.3f} median {b:.3f} ms")
