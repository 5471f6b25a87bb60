#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpc -- python $GRAFT_REPO_ROOT/scripts/dev/regime_cats_only.py $1 2>&1 | grep -v "amdgpu\|simple_timer" | tail -4
f=$(find /tmp/rpc -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:10]:
    print(f"   {r['Name'].split('(')[0].replace('void ', '')[:70]:72s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
PY
