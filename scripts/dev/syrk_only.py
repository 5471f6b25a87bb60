import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = 10_000_000
dm = synth.dense_block(n, 128, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
_lib.call("tm_profile_enable", 1)
ts = []
for _ in range(4):
    dm._sandwich_dev(d, None, None)
    ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
torch.cuda.synchronize()
print("syrk f64 10Mx128:", min(ts), "ms")
