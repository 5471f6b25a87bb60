"""the cfg4 step panel by panel, every block product of a panel back to back, so that the
second and third reader of a block find it in the 256 MiB Infinity Cache.  No new kernel: the panels are device row
slices of the 10M-row matrix (SplitMatrix.__getitem__, twins per slice), the result is the sum of their sandwiches.

    P=196608 python scripts/dev/panels_step0.py            # wall time of a pass (HIP events) vs the full step
    rocprofv3 --kernel-trace --stats ... -- python scripts/dev/panels_step0.py     # per-kernel sums over the pass

P = rows per panel (0 = the unpanelled step).  PASSES = timed passes.  Prints one line."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tabmat_amd import synth  # noqa: E402

N = int(os.environ.get("N", 10_000_000))
P = int(os.environ.get("P", 0))
PASSES = int(os.environ.get("PASSES", 5))
X = synth.mixed_split(N, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(N, dtype=torch.float64, device="cuda")
if P <= 0:
    panels = [(0, N, X)]
else:
    panels = [(a, min(a + P, N), X[a:min(a + P, N)]) for a in range(0, N, P)]
ds = [d[a:b].contiguous() for a, b, _ in panels]


def one_pass():
    out = None
    for (a, b, part), dd in zip(panels, ds):
        r = part._sandwich_dev(dd, None, None)
        out = r if out is None else out.add_(r)
    return out


ref = None
for _ in range(2):          # twins, workspaces, int8 history
    ref = one_pass()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(PASSES):
    e0.record()
    one_pass()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
full = X._sandwich_dev(d, None, None) if P > 0 and os.environ.get("CHECK", "1") == "1" else None
err = ""
if full is not None:
    err = f" max|panels - full| / max|full| = {float((ref - full).abs().max() / full.abs().max()):.2e}"
print(f"P={P} panels={len(panels)} pass ms: min {min(ts):.3f} mean {sum(ts) / len(ts):.3f}{err}", flush=True)
