#!/bin/bash
# kernel-time breakdown of any dev script: bash scripts/dev/prof_script.sh scripts/dev/x.py [args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpx; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpx -- python $GRAFT_REPO_ROOT/"$@" 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -6
f=$(find /tmp/rpx -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    n = int(r["Calls"])
    print(f"   {r['Name'].split('(')[0].replace('void ', '')[:78]:80s} calls {n:4d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
PY
