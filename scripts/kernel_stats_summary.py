"""Condense a rocprofv3 --kernel-trace --stats CSV to the tabmat (tmh::) kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "tmh::" in r["Name"])
print("# per-kernel durations of one rocprofv3 --kernel-trace --stats run of bench.py (tabmat kernels only)")
print(f"{'kernel':60s} {'calls':>6s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'total_ms':>10s} {'%tmh':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "tmh::" not in r["Name"]:
        continue
    name = r["Name"].split("tmh::")[1].split("(")[0][:60]
    print(f"{name:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e6:10.4f} {float(r['MinNs'])/1e6:10.4f} "
          f"{float(r['MaxNs'])/1e6:10.4f} {float(r['TotalDurationNs'])/1e6:10.3f} "
          f"{100*float(r['TotalDurationNs'])/tot:6.2f}")
