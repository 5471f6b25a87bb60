"""Phase stamps of the int8 syrk's chunk loop (I8_TRACE build: scripts/dev/build_variant.sh i8_trace syrk_i8.hip -DI8_TRACE)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth
from tabmat_amd.ext import dense as xd
n = int(os.environ.get("N", 10_000_000))
dm = synth.dense_block(n, 128, torch.float64, 3)
d = torch.rand(n, dtype=torch.float64, device="cuda")
Xd = dm._dev_c()
cmax = Xd.as_2d().abs().amax(dim=0).contiguous()
for _ in range(3):
    out = xd.dense_sandwich_i8(Xd, d, cmax)
t = out.flatten()[:128].cpu().numpy().reshape(4, 4, 8)        # [iteration][wave][stamp]
names = ["top", "A", "frags+27", "B", "G0-3", "vm+pub", "C", "G4-7"]
for it in range(4):
    for w in range(4):
        r = t[it, w]
        nxt = t[it + 1, w, 0] if it < 3 else float("nan")
        seg = [r[k + 1] - r[k] for k in range(7)] + [nxt - r[7]]
        print(f"it {it} wave {w}: start {r[0]:8.0f} | " + "  ".join(f"{nm}:{v:6.0f}" for nm, v in zip(
            ["->A", "A->fr", "fr->B", "B->G3", "vmwait", "pub->C", "C->G7", "vm->top"], seg)))
