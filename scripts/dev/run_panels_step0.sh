#!/bin/bash
# per-kernel time of ONE pass (sum over the panels) for each panel height, from
# rocprofv3 --kernel-trace --stats over 7 passes (2 warm + 5 timed), next to the unpanelled step (P=0).
#   bash scripts/dev/run_panels_step0.sh "0 131072 196608 262144 524288 1048576" > gpurun_out/r6_panels_step0.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for P in ${1:-0 131072 196608 262144 524288 1048576}; do
  rm -rf /tmp/rpp
  CHECK=0 P=$P PASSES=5 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpp -- python $R/scripts/dev/panels_step0.py 2>&1 | grep "^P="
  f=$(find /tmp/rpp -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tmh::" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 7e6
print(f"   sum of tabmat kernels per pass: {tot:8.3f} ms")
small = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 7e6
    if ms < 0.15:
        small += ms
        continue
    print(f"   {r['Name'].split('tmh::')[1].split('(')[0][:60]:62s} calls/pass {int(r['Calls']) / 7:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  per pass {ms:8.3f} ms")
print(f"   {'(kernels below 0.15 ms per pass)':62s} {'':19s} {'':17s}  per pass {small:8.3f} ms")
PY
done
