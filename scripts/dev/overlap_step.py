# NOTE (round 4): the guest-stream / stream-fan hooks this harness drives (split_matrix.OVERLAP, _StreamFan) were
# removed from the product path after rounds 2-3 measured them at 0.0-0.3 ms (profiles/r3_coresidency.txt);
# it runs against the tree of commit 60e1c99.
"""cfg4 step with / without the guest-stream overlap of the dense syrk, knob sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd.split_matrix as sm
from tabmat_amd import _lib, synth

n = int(os.environ.get("N", 10_000_000))
mat = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")


def step_ms(reps=8):
    for _ in range(2):
        out = mat.sandwich(d)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = mat.sandwich(d)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts) // 2], out


sm.OVERLAP = False
t0, m0, ref = step_ms()
print(f"no overlap            : min {t0:.3f} med {m0:.3f} ms")
sm.OVERLAP = True
for knobs in ({"k2_waves": 12, "catdense_waves": 12, "catsparse_waves": 12},
              {"k2_waves": 16, "catdense_waves": 12, "catsparse_waves": 12},
              {"k2_waves": 12, "catdense_waves": 16, "catsparse_waves": 16},
              {"k2_waves": 16, "catdense_waves": 16, "catsparse_waves": 16},
              {"k2_waves": 12, "catdense_waves": 12, "catsparse_waves": 16},
              {"k2_waves": 8, "catdense_waves": 12, "catsparse_waves": 12},
              {"k2_waves": 12, "catdense_waves": 8, "catsparse_waves": 12}):
    sm.OVERLAP_KNOBS = knobs
    for g in (256, 512, 768):
        _lib.call("tm_tune_set", b"co_grid", g)
        t, m, out = step_ms()
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        print(f"overlap {knobs} grid {g}: min {t:.3f} med {m:.3f} ms  (vs no-overlap result {err:.1e})")
