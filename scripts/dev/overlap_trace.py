# NOTE (round 4): the guest-stream / stream-fan hooks this harness drives (split_matrix.OVERLAP, _StreamFan) were
# removed from the product path after rounds 2-3 measured them at 0.0-0.3 ms (profiles/r3_coresidency.txt);
# it runs against the tree of commit 60e1c99.
"""One cfg4 step under rocprofv3 --kernel-trace: start / end of every kernel of the last step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tabmat_amd.split_matrix as sm
from tabmat_amd import _lib, synth

n = int(os.environ.get("N", 10_000_000))
mat = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
sm.OVERLAP = os.environ.get("OVERLAP", "1") == "1"
_lib.call("tm_tune_set", b"co_grid", int(os.environ.get("GRID", 256)))
for _ in range(4):
    mat.sandwich(d)
torch.cuda.synchronize()
