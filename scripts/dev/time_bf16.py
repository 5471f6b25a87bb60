"""cfg2 (10M x 256 f32): the bf16x3 syrk against the f32-MFMA syrk, kernel times from the library's event pair."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
n = int(os.environ.get("N", 10_000_000))
dm = synth.dense_block(n, 256, torch.float32, 1)
d = torch.rand(n, dtype=torch.float32, device="cuda")
_lib.call("tm_profile_enable", 1)
res = {}
for name, knob in (("bf16x3", 1), ("f32 mfma", 0)):
    _lib.call("tm_tune_set", b"syrk_bf16", knob)
    for g in ((256, 512) if knob else (0,)):
        if knob:
            _lib.call("tm_tune_set", b"bx_grid", g)
        ts = []
        for _ in range(5):
            out = dm._sandwich_dev(d, None, None)
            ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
        res[name] = out
        print(f"{name:9s} grid {g:4d}: min {min(ts):.3f} ms  median {sorted(ts)[2]:.3f}")
a, b = res["bf16x3"].double(), res["f32 mfma"].double()
print("max |bf16x3 - f32| / max|f32| =", ((a - b).abs().max() / b.abs().max()).item())
