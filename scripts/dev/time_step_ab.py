"""cfg4 step (wall time of SplitMatrix.sandwich, device vector) + the main-kernel times of its ops: for same-box A/B runs of
two libraries (TABMAT_AMD_LIB)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from tabmat_amd import synth
n = 10_000_000
mat = synth.mixed_split(n)
mat.to_device()
d = torch.rand(n, dtype=torch.float64, device="cuda")
for _ in range(3): mat.sandwich(d)
torch.cuda.synchronize(); ts = []
for _ in range(12):
    t0 = time.perf_counter(); mat.sandwich(d); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
bd = bench.kernel_breakdown(mat, d, reps=4)
print(f"step min {min(ts):.3f} median {sorted(ts)[6]:.3f} ms   " + " ".join(f"{k}={v:.3f}" for k, v in sorted(bd.items(), key=lambda kv: -kv[1]) if v > 0.3), flush=True)
