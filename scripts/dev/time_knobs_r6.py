"""cfg4 step against the knobs that the round-6 kernels may have moved (tm_tune_set): waves / rounds of the staged
categorical x sparse kernel, K3's workgroup rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = 10_000_000
mat = synth.mixed_split(n)
mat.to_device()
d = torch.rand(n, dtype=torch.float64, device="cuda")
def t():
    for _ in range(3): mat.sandwich(d)
    torch.cuda.synchronize(); ts = []
    for _ in range(8):
        t0 = time.perf_counter(); mat.sandwich(d); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[4]
for key, vals in ((b"catsparse_waves", (16, 12, 8, 16)), (b"catsparse_rounds", (1, 2, 1)), (b"ent_rounds", (1, 2, 1)),
                  (b"catsparse_staged", (1, 0, 1))):
    for v in vals:
        _lib.call("tm_tune_set", key, v)
        a, b = t()
        print(f"{key.decode()} {v}: step min {a:.3f} median {b:.3f} ms", flush=True)
    _lib.call("tm_tune_set", key, -2**63)
