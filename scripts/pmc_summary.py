"""Average (and min / max over dispatches) of the rocprofv3 --pmc counters per tabmat kernel (kernels whose name
holds `tmh::`, or the substring given as the second argument)."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    tag = sys.argv[2] if len(sys.argv) > 2 else "tmh::"
    if tag not in name:
        continue
    key = name.split("tmh::")[1][:44] if "tmh::" in name else name[:44]
    vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    print(k)
    for c, xs in v.items():
        print(f"    {c:28s} {sum(xs) / len(xs):.4g}   (min {min(xs):.4g}  max {max(xs):.4g}  n={len(xs)})")
