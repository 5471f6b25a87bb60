"""Kernels between the last two launches of a marker kernel in a rocprofv3 kernel-trace csv."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
marker = sys.argv[2]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
tot = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
print(f"span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, {len(seg)} launches")
