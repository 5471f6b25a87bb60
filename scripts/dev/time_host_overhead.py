"""Host-side cost of one SplitMatrix.sandwich call (enqueue only, no sync) and the end-to-end
latency for small matrices, eager vs HIP graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
for n in (10_000, 100_000, 1_000_000):
    X = synth.mixed_split(n)
    d = torch.rand(n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        X.sandwich(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        X.sandwich(d)
    t_enq = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        X.sandwich(d)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    g = X.sandwich_graph(d)
    for _ in range(3):
        g(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g(d)
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 20
    print(f"n = {n:8d}: host enqueue {t_enq * 1e3:6.2f} ms   eager end-to-end {t_all * 1e3:6.2f} ms   graph replay {t_graph * 1e3:6.2f} ms", flush=True)
