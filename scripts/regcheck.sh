#!/bin/bash
# usage: scripts/regcheck.sh <file.hip> [kernel-name-filter]   -- per-kernel register / scratch use (gfx950)
src=/root/repo/tabmat_amd/csrc/$1
mkdir -p /tmp/t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -munsafe-fp-atomics -O3 -std=c++17 --cuda-device-only -S -o /tmp/t/${1%.hip}.s $src 2>&1 | grep -E "error" -A6
python3 - "$1" "${2:-}" <<'PY'
import re, sys
txt = open(f"/tmp/t/{sys.argv[1][:-4]}.s").read()
for blk in txt.split("  - .agpr_count:")[1:]:
    f = {k: v for k, v in re.findall(r"\.(agpr_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_count|name):\s+(\S+)", "  - .agpr_count:" + blk)}
    if sys.argv[2] in f.get("name", ""):
        print(f"{f.get('name','?')[:70]:70s} vgpr {f.get('vgpr_count')} agpr {f.get('agpr_count')} sgpr {f.get('sgpr_count')} spill {f.get('vgpr_spill_count')} scratch {f.get('private_segment_fixed_size')}")
PY
