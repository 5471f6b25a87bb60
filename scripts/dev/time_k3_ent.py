"""K3: the entry-list kernel (round 4, csrc/sparse_ent.hip) against the lane-group kernel; small-size
check against a dense product first.  usage: python scripts/dev/time_k3_ent.py [rows] [f64|f32] [density]"""
import os, sys, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext._types import SlabEnt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05


def timed(fn, reps=4):
    _lib.call("tm_profile_enable", 1)
    ts = []
    for _ in range(reps):
        out = fn()
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    _lib.call("tm_profile_enable", 0)
    return min(ts), out


# ---- small check vs dense algebra (ragged last slab, zeros in d, an empty column group) ----
for (ns, ms, kb, dn) in ((1000, 40, 128, 0.3), (5003, 100, 136, 0.05), (70, 16, 128, 0.9), (20011, 512, 256, 0.02)):
    for dts in (torch.float64, torch.float32):
        dmx = synth.dense_block(ns, kb, dts, 3)
        smx = synth.sparse_block(ns, ms, dn, dts, 7)
        dd = torch.rand(ns, dtype=dts, device="cuda")
        dd[::7] = 0
        ent = SlabEnt.from_csr(smx._dev())
        out, cs = xs.csr_dense_sandwich_ent(ent, dmx._dev_c(), dd, want_colsum=True)
        out2 = xs.csr_dense_sandwich_ent(ent, dmx._dev_c(), dd)
        A = torch.tensor(smx.array_csc.toarray(), device="cuda", dtype=torch.float64)
        Bm = dmx._dev_c().as_2d().to(torch.float64)
        ref = A.T @ (dd.to(torch.float64)[:, None] * Bm)
        refc = A.T @ dd.to(torch.float64)
        e = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        e2 = ((out2.double() - ref).abs().max() / ref.abs().max()).item()
        ec = ((cs.double() - refc).abs().max() / refc.abs().max()).item()
        print(f"check n={ns} m={ms} k={kb} dens={dn} {dts}: rel.err {e:.2e} / {e2:.2e}  colsum {ec:.2e}", flush=True)

dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, dens, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
lg = sm._lg()
if lg is not None:
    t0, ref = timed(lambda: xs.csr_dense_sandwich_lg(lg, Bd, d))
    print(f"lg: {t0:.3f} ms", flush=True)
else:
    ell = sm._ell(wide=True)
    t0, ref = timed(lambda: xs.csr_dense_sandwich_ell(ell, Bd, d))
    print(f"ellw: {t0:.3f} ms", flush=True)
torch.cuda.synchronize(); t = time.time()
ent = SlabEnt.from_csr(sm._dev())
torch.cuda.synchronize()
nnz = sm._dev().data.numel()
print(f"ent twin: built in {time.time()-t:.2f} s, slots {(ent.vals.numel()-ent.SLACK)/nnz:.3f}x nnz, "
      f"{ent.nbytes()/1e9:.2f} GB", flush=True)
t1, out = timed(lambda: xs.csr_dense_sandwich_ent(ent, Bd, d))
err = ((out - ref).abs().max() / ref.abs().max()).item()
print(f"ent: {t1:.3f} ms  rel.diff vs lg {err:.2e}", flush=True)
t2, (out, cs) = timed(lambda: xs.csr_dense_sandwich_ent(ent, Bd, d, want_colsum=True))
print(f"ent + colsum: {t2:.3f} ms", flush=True)
