"""Section cycle counters of the entry-list K3 kernel (build with -DEN_PROF: scripts/dev/build_variant.sh
ent_prof sparse_ent.hip "-Wno-inline-asm -DEN_PROF"; run with TABMAT_AMD_LIB=tabmat_amd/_abl/libtabmat_ent_prof.so)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
from tabmat_amd.ext._types import SlabEnt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
dt = torch.float64
dm = synth.dense_block(n, 128, dt, 3)
sm = synth.sparse_block(n, 512, dens, dt, 1003)
d = torch.rand(n, dtype=dt, device="cuda")
Bd = dm._dev_c()
ent = SlabEnt.from_csr(sm._dev())
prof = torch.zeros(256 * 16 * 8, dtype=torch.int64, device="cuda")
for _ in range(2):
    xs.csr_dense_sandwich_ent(ent, Bd, d)
_lib.call("tm_tune_set", b"ent_prof", prof.data_ptr())
_lib.call("tm_profile_enable", 1)
xs.csr_dense_sandwich_ent(ent, Bd, d)
ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms))
torch.cuda.synchronize()
p = prof.view(-1, 16, 8).double().cpu()
p = p[p[:, 0, 7] > 0]
names = ["stage A (requests)", "copy pieces", "batch", "stage B (fold)", "barrier", "batches", "total", "slabs"]
print(f"kernel {ms.value:.3f} ms; {p.shape[0]} workgroups; per wave means (cycles of s_memtime = 100 MHz? see total):")
for k, nm in enumerate(names):
    print(f"  {nm:22s} mean {p[:, :, k].mean():14.1f}   min {p[:, :, k].min():12.1f}  max {p[:, :, k].max():12.1f}")
tot = p[:, :, 6].mean()
for k in range(5):
    print(f"  {names[k]:22s} {100 * p[:, :, k].mean() / tot:5.1f} % of the loop")
print(f"  per batch: batch section {p[:, :, 2].sum() / p[:, :, 5].sum():.1f}, A {p[:, :, 0].sum() / p[:, :, 5].sum():.1f}, "
      f"B {p[:, :, 3].sum() / p[:, :, 5].sum():.1f} ticks; batches per slab {p[:, :, 5].sum() / p[:, :, 7].sum():.2f}; "
      f"barrier per slab {p[:, :, 4].sum() / p[:, :, 7].sum():.1f}")
