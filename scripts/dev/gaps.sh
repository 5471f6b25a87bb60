#!/bin/bash
# per-dispatch timeline of one sandwich step: kernel, duration, gap to the previous kernel
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp && mkdir -p /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-traffic --no-cpu-baseline > /tmp/gp/out.txt 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# one timed step: from the 4th multi_cat_dense_wide dispatch to the 5th
starts = [i for i, k in enumerate(ks) if "multi_cat_dense_wide" in k[0]]
lo, hi = starts[3], starts[4]
prev = None
tot_gap = 0
n = 0
for name, s, e in ks[lo:hi]:
    gap = (s - prev) / 1e3 if prev else 0
    tot_gap += max(gap, 0)
    n += 1
    short = name.split("(")[0].replace("void tmh::", "")[:70]
    print(f"{short:72s} dur {(e - s) / 1e3:9.1f} us   gap {gap:8.1f} us")
    prev = e
print(f"{n} dispatches, step span {(ks[hi][1] - ks[lo][1]) / 1e3:.1f} us, sum of gaps {tot_gap:.1f} us")
PY
