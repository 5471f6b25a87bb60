"""K2b with the cyclic row-range deal (TABMAT_K2B_CYCLIC / _NWG / _ROUNDS) against the contiguous one; checks the
result against the contiguous build's."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = int(os.environ.get("N", 10_000_000))
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
sm = X.matrices[1]
_lib.call("tm_profile_enable", 1)
ref = None
for cfg in sys.argv[1:] or ["0:512"]:
    cyc, nwg = cfg.split(":")[:2]
    kw = dict(cyclic=int(cyc), n_wg=int(nwg))
    sm._dev()._pb = None
    sm._dev().pair_blocks(**kw)
    ts = []
    for _ in range(5):
        out = sm._sandwich_dev(d, None, None)
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max() / ref.abs().max())
    import time
    for _ in range(3):
        X.sandwich(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        X.sandwich(d)
    torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 20 * 1e3
    print(f"cyclic {cyc:>6s} n_wg {nwg:>4s}: K2b {min(ts):.3f} ms   step {step:.3f} ms   max rel diff vs first {err:.2e}", flush=True)
