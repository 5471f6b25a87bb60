"""K2 at cfg4 size: chunked kernel vs block-list kernel (library event pair around the main kernel)."""
import os, sys, ctypes as C, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabmat_amd import synth, _lib
from tabmat_amd.ext import sparse as xs
n = int(os.environ.get("N", 10_000_000))
sm = synth.sparse_block(n, 512, 0.05, torch.float64, 1003)
d = torch.rand(n, dtype=torch.float64, device="cuda")
A = sm._dev()
A.chunk_major()
torch.cuda.synchronize(); t0 = time.perf_counter()
NW = int(os.environ.get("K2B_WAVES", 16))
_lib.call("tm_tune_set", b"k2b_waves", NW)
blocks, wg_tab, max_nb = A.pair_blocks(int(os.environ.get("NWG", 512)), NW)
torch.cuda.synchronize()
print(f"block list: {blocks.shape[0]} blocks ({blocks.shape[0] / n / 10:.3f} per row and tile), "
      f"{wg_tab.shape[0]} workgroups, max {max_nb} per tile, built in {time.perf_counter() - t0:.2f} s, "
      f"{blocks.numel() * 4 / 1e9:.2f} GB")
_lib.call("tm_profile_enable", 1)
for name, fn in (("chunked", xs.sparse_sandwich_chunked), ("blocks", xs.sparse_sandwich_blocks)):
    ts = []
    for _ in range(6):
        out = fn(A, d)
        ms = C.c_float(0); _lib.call("tm_profile_last_ms", C.byref(ms)); ts.append(ms.value)
    print(f"K2 {name:8s} min {min(ts):.3f} ms  median {sorted(ts)[3]:.3f}  checksum {out.sum().item():.10e}")
