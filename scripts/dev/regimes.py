"""SplitMatrix.sandwich across regimes other than cfg4: density, widths, dtype, layout, level
counts -- looking for performance cliffs outside the tuned shape.  2M rows each.
usage: python scripts/dev/regimes.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
N = 2_000_000
for kv in os.environ.get("TUNE", "").split(","):          # e.g. TUNE=catdense_waves=8
    if "=" in kv:
        from tabmat_amd import _lib
        _lib.call("tm_tune_set", kv.split("=")[0].encode(), int(kv.split("=")[1]))
CASES = [
    ("cfg4 shape (dense128 + sp512@5% + cats 256/96/32)", dict()),
    ("float32", dict(dtype=torch.float32)),
    ("density 1 %", dict(density=0.01)),
    ("density 0.2 %", dict(density=0.002)),
    ("density 20 %", dict(density=0.20)),
    ("sparse 2048 cols @ 1.25 %", dict(k_sparse=2048, density=0.0125)),
    ("sparse 4096 cols @ 0.2 %", dict(k_sparse=4096, density=0.002)),
    ("sparse 8192 cols @ 0.05 %", dict(k_sparse=8192, density=0.0005)),
    ("sparse 100 cols @ 5 %", dict(k_sparse=100)),
    ("dense 64 cols", dict(k_dense=64)),
    ("dense 256 cols", dict(k_dense=256)),
    ("dense 50 cols (unaligned)", dict(k_dense=50)),
    ("cats 10000 / 500", dict(cats=(10000, 500))),
    ("cats 5 x 20", dict(cats=(20, 20, 20, 20, 20))),
    ("cats 12 x 30", dict(cats=(30,) * 12)),
    ("cats 24 x 10, dense 32, sparse 128", dict(cats=(10,) * 24, k_dense=32, k_sparse=128)),
    ("no categoricals", dict(cats=())),
    ("no sparse block", dict(k_sparse=0)),
]
if len(sys.argv) > 1:          # a single case (by index), e.g. under rocprofv3 --kernel-trace --stats
    CASES = [CASES[int(sys.argv[1])]]
print(f"{'case':52s} {'ms':>8s} {'GB/s':>8s} {'GFLOP/s':>9s}")
for name, kw in CASES:
    try:
        if kw.get("k_sparse", 1) == 0:
            kw = dict(kw); kw.pop("k_sparse")
            dt = kw.get("dtype", torch.float64)
            import numpy as np
            from tabmat_amd.split_matrix import SplitMatrix
            blocks = [synth.dense_block(N, 128, dt, 3)] + [synth.cat_block(N, c, 2000 + i) for i, c in enumerate((256, 96, 32))]
            X = SplitMatrix(blocks)
        else:
            X = synth.mixed_split(N, **kw)
        d = torch.rand(N, dtype=kw.get("dtype", torch.float64), device="cuda")
        for _ in range(2):
            X.sandwich(d)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); X.sandwich(d); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ms = min(ts) * 1e3
        print(f"{name:52s} {ms:8.3f} {synth.algorithmic_bytes(X) / ms / 1e6:8.0f} {synth.algorithmic_flops(X) / ms / 1e6:9.0f}", flush=True)
    except Exception as e:  # noqa
        import traceback
        print(f"{name:52s} FAILED: {type(e).__name__}: {str(e)[:100]}", flush=True)
        print(''.join(traceback.format_exc().splitlines(True)[-8:]), flush=True)
    X = None
    torch.cuda.empty_cache()
