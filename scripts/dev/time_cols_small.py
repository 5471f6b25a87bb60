"""cfg4 recipe at 2M rows: sandwich with 5 % of the columns, for rocprofv3 (scripts/dev/prof_script.sh)."""
import os, sys, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
N = int(os.environ.get("TM_ROWS", "2000000"))
X = synth.mixed_split(N)
p = X.shape[1]
d = torch.rand(N, dtype=torch.float64, device="cuda")
share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
cols = np.sort(np.random.default_rng(0).choice(p, int(share * p), replace=False))
for _ in range(7):
    X.sandwich(d, cols=cols)
torch.cuda.synchronize()
