"""does a block product run faster when the previous product of the SAME row panel has just
pulled its operand through the 256 MiB Infinity Cache?  Same launches in two orders, so the fixed costs of a small
launch (tile set-up / flush, ramp, tail) cancel:

  kernel-major ("cold"):  for op: for panel: op(panel)        -- an operand was last touched a whole pass ago
  panel-major  ("warm"):  for panel: for op: op(panel)        -- the second / third reader follows the first at once
  repeat       ("hot"):   for panel: for op: op(panel); op(panel) -- second call timed: everything it reads was read
                                                                  by the very same kernel a moment ago (upper bound)

Main-kernel time of every op through tm_profile (HIP events around the op's main kernel), summed over the panels.
Order of the ops = the order a panel-major step would use: dense self (first reader of the dense block), categorical x
dense, sparse x dense (second / third readers of the dense block, first of the entry twin), categorical x sparse
(second reader of the entry twin), sparse self (its own twin)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tabmat_amd import _lib, synth  # noqa: E402

N = int(os.environ.get("N", 10_000_000))
ORDER = ["dense0.self", "allcats_x_dense0", "dense0xsparse1", "allcats_x_sparse1", "sparse1.self"]
X = synth.mixed_split(N, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
d = torch.rand(N, dtype=torch.float64, device="cuda")
X.sandwich(d)


def last_ms():
    ms = C.c_float(0)
    _lib.call("tm_profile_last_ms", C.byref(ms))
    return ms.value


for P in [int(x) for x in os.environ.get("PS", "65536 131072 196608 262144").split()]:
    panels = []
    for a in range(0, N - P + 1, P):
        part = X[a:a + P]
        dd = d[a:a + P].contiguous()
        part.sandwich(dd)                                   # twins of the slice
        ops = dict(bench.kernel_ops(part, dd))
        panels.append([ops[k] for k in ORDER])
        if len(panels) * P >= 4_000_000:                    # ~4M rows of panels: several times the cache
            break
    _lib.call("tm_profile_enable", 1)
    res = {}
    for mode in ("cold", "warm", "hot"):
        tot = [0.0] * len(ORDER)
        for rep in range(3):
            acc = [0.0] * len(ORDER)
            if mode == "cold":
                for k in range(len(ORDER)):
                    for ops in panels:
                        ops[k]()
                        acc[k] += last_ms()
            else:
                for ops in panels:
                    for k in range(len(ORDER)):
                        ops[k]()
                        if mode == "hot":
                            ops[k]()
                        acc[k] += last_ms()
            if rep:
                tot = [t + a for t, a in zip(tot, acc)]
        scale = N / (len(panels) * P) / 2.0                 # per 10M-row pass
        res[mode] = [t * scale for t in tot]
    _lib.call("tm_profile_enable", 0)
    print(f"P={P} ({len(panels)} panels timed, scaled to {N} rows): main-kernel ms per pass", flush=True)
    print(f"   {'op':20s} {'cold':>8s} {'warm':>8s} {'hot':>8s}")
    for k, name in enumerate(ORDER):
        print(f"   {name:20s} {res['cold'][k]:8.3f} {res['warm'][k]:8.3f} {res['hot'][k]:8.3f}")
    print(f"   {'sum':20s} {sum(res['cold']):8.3f} {sum(res['warm']):8.3f} {sum(res['hot']):8.3f}", flush=True)
    del panels
