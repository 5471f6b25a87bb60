"""Timeline of the LAST step in a rocprofv3 kernel-trace csv: name, start offset, duration (ms)."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last fill/zero of the out buffer: take everything after the last
# syrk_co_kernel start minus a margin -> simply the last 40 kernels
big = [r for r in rows if "syrk_co_kernel" in r["Kernel_Name"]]
t_ref = int(big[-1]["Start_Timestamp"]) - 200_000
last = [r for r in rows if int(r["Start_Timestamp"]) >= t_ref]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) > 20_000:
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e6:6.3f} ms)  {r['Kernel_Name'][:90]}")
print("span of the step: %.3f ms" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
