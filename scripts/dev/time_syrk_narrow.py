"""DenseMatrix.sandwich for narrow blocks (the block T of a narrow column selection): ms and GB/s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
N = int(os.environ.get("TM_ROWS", "10000000"))
for dt in (torch.float64, torch.float32):
    for k in (8, 16, 32, 48, 64, 96, 128):
        X = synth.dense_block(N, k, dt, seed=1)
        d = torch.rand(N, dtype=dt, device="cuda")
        for _ in range(3):
            X.sandwich(d)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); X.sandwich(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ms = min(ts) * 1e3
        gb = N * k * X._dev().buf.element_size() / 1e9
        print(f"{str(dt)[6:]:8s} k = {k:4d}: {ms:7.3f} ms  {gb / ms * 1e3:7.0f} GB/s", flush=True)
        del X
