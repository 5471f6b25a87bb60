"""Time SplitMatrix.matvec / transpose_matvec (device in/out) on the cfg4 workload."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabmat_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
X = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
v = torch.rand(X.shape[1], dtype=torch.float64, device="cuda")
w = torch.rand(n, dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print(f"matvec            {t(lambda: X.matvec(v)):8.3f} ms")
print(f"transpose_matvec  {t(lambda: X.transpose_matvec(w)):8.3f} ms")
for i, m in enumerate(X.matrices):
    vi = v[:m.shape[1]].contiguous()
    name = type(m).__name__
    if name == "CategoricalMatrix":
        print(f"  {name}{i} matvec {t(lambda: m._matvec_dev(vi, None, None)):7.3f}  rmatvec {t(lambda: m._transpose_matvec_dev(w, None, None, torch.zeros(m.shape[1], dtype=torch.float64, device='cuda'))):7.3f}")
    else:
        print(f"  {name}{i} matvec {t(lambda: m._matvec_dev(vi, None, None, None, False)):7.3f}  rmatvec {t(lambda: m._matvec_dev(w, None, None, None, True)):7.3f}")
