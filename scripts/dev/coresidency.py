"""Co-residency experiment: the MFMA-bound dense syrk (guest kernel, syrk_co.hip)
beside each of the other big kernels of the cfg4 step on the same CUs.  Prints every kernel alone,
every pair on two streams, and -- from the per-workgroup placement log ("wg_log") -- how many guest
workgroups ran on a CU at the same time as a workgroup of the partner."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import _lib, synth
from tabmat_amd.ext import dense as xd
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext import split as xsplit

n = int(os.environ.get("N", 10_000_000))
mat = synth.mixed_split(n, 128, 512, (256, 96, 32), 0.05, torch.float64, 3)
dm, sm = mat.matrices[0], mat.matrices[1]
cats = [(m._dev(), m.shape[1], m.drop_first) for m in mat.matrices[2:]]
d = torch.rand(n, dtype=torch.float64, device="cuda")
Xd = dm._dev_c()
A = sm._dev()
A.chunk_major()
slab = sm._slab()
lg = sm._lg()


def tune(k, v):
    _lib.call("tm_tune_set", k.encode(), int(v))


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


ref = xd.dense_sandwich(Xd, d, None, None)
out, cs = xd.dense_sandwich_co(Xd, d, want_colsum=True)
cs_ref = xd.dense_rmatvec(Xd, d, None, None)
torch.cuda.synchronize()
print("syrk_co rel.err", ((out - ref).abs().max() / ref.abs().max()).item(),
      "colsum rel.err", ((cs - cs_ref).abs().max() / cs_ref.abs().max()).item(),
      "symmetric", bool((out == out.T).all().item()))

co = lambda: xd.dense_sandwich_co(Xd, d)
partners = {
    "K2": (lambda: xs.sparse_sandwich_chunked(A, d), "k2_waves"),
    "catdense": (lambda: xsplit.multi_cat_dense_sandwich(cats, d, Xd), "catdense_waves"),
    "catsparse": (lambda: xsplit.multi_cat_sparse_sandwich(cats, d, slab), "catsparse_waves"),
    "K3": (lambda: xs.csr_dense_sandwich_lg(lg, Xd, d), None),
}
print("syrk plain alone: %.3f ms" % timeit(lambda: xd.dense_sandwich(Xd, d, None, None)))
alone_co = {}
for g in (256, 512, 768):
    tune("co_grid", g)
    alone_co[g] = timeit(co)
    print(f"syrk_co grid {g} alone: {alone_co[g]:.3f} ms")

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sB):
    co()
torch.cuda.synchronize()
LOGCAP = 8192
logbuf = torch.zeros((LOGCAP + 1, 4), dtype=torch.int64, device="cuda")


def pair(first, second, gap_us=0):
    main = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    sA.wait_event(a)
    sB.wait_event(a)
    with torch.cuda.stream(sA):
        first()
    with torch.cuda.stream(sB):
        if gap_us:
            torch.cuda._sleep(int(gap_us * 2100))
        second()
    main.wait_stream(sA)
    main.wait_stream(sB)
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def cu_key(hw):
    xcc = (hw >> 32) & 0xF
    h = hw & 0xFFFFFFFF
    return xcc * 10000 + ((h >> 13) & 0x7) * 1000 + ((h >> 12) & 1) * 100 + ((h >> 8) & 0xF)


def placement(tagp):
    L = logbuf.cpu().numpy()
    cnt = int(L[0, 0])
    E = L[1:1 + min(cnt, LOGCAP)]
    G, P = E[E[:, 3] == 1], E[E[:, 3] == tagp]
    if len(G) == 0 or len(P) == 0:
        return "no log"
    byc = {}
    for hw, a, b, _ in P:
        byc.setdefault(cu_key(int(hw)), []).append((a, b))
    ov_t, g_t, nov = 0, 0, 0
    for hw, a, b, _ in G:
        o = sum(max(0, min(b, kb) - max(a, ka)) for ka, kb in byc.get(cu_key(int(hw)), []))
        ov_t += min(o, b - a)
        g_t += b - a
        nov += o > 0
    pspan = (P[:, 2].max() - P[:, 1].min()) / 100.0
    gspan = (G[:, 2].max() - G[:, 1].min()) / 100.0
    return (f"partner wgs {len(P)} on {len(byc)} CUs (span {pspan:.0f} us), guest wgs {len(G)} "
            f"(span {gspan:.0f} us); guests that shared a CU with the partner: {nov}; share of "
            f"guest wg-time beside the partner: {ov_t / max(g_t, 1):.2f}")


TAG = {"K2": 2, "catdense": 3, "catsparse": 4, "K3": 5}
for name, (fn, knob) in partners.items():
    with torch.cuda.stream(sA):
        fn()
    torch.cuda.synchronize()
    for w in ((16, 12, 8) if knob else (16,)):
        if knob:
            tune(knob, w)
        t_alone = timeit(fn)
        for g in (256, 512):
            tune("co_grid", g)
            ts = [pair(fn, co) for _ in range(5)]
            logbuf.zero_()
            logbuf[0, 1] = LOGCAP
            tune("wg_log", logbuf.data_ptr())
            pair(fn, co)
            tune("wg_log", 0)
            print(f"{name:9s} {w:2d} waves alone {t_alone:.3f} ms | + syrk_co grid {g} (alone {alone_co[g]:.3f}): "
                  f"pair min {min(ts):.3f} ms, serial sum {t_alone + alone_co[g]:.3f} | {placement(TAG[name])}")
    if knob:
        tune(knob, 16)
