#!/bin/bash
# kernel-time breakdown of single regimes (indices into scripts/dev/regimes.py CASES)
cd /tmp && export TMPDIR=/tmp
for i in "$@"; do
  rm -rf /tmp/rp$i; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$i -- python $GRAFT_REPO_ROOT/scripts/dev/regimes.py $i 2>&1 | grep -v amdgpu.ids | tail -2
  f=$(find /tmp/rp$i -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    n = int(r["Calls"])
    print(f"   {r['Name'].split('(')[0].replace('void ', '')[:78]:80s} calls {n:4d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
PY
done
