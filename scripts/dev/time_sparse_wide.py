import sys, time, torch
sys.path.insert(0, "/root/repo")
from tabmat_amd import synth
m = synth.reference_design("sparse_wide")
d = torch.rand(m.shape[0], dtype=torch.float64, device="cuda")
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        t0=time.perf_counter(); r=fn(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    return min(ts), r
a, ra = t(lambda: m.sandwich(d))
m._direct_pays = True
b, rb = t(lambda: m.sandwich(d))
print(f"generic {a:.2f} ms   direct {b:.2f} ms   max diff {(ra-rb).abs().max().item():.3e} of {ra.abs().max().item():.3e}")
