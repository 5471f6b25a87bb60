"""Average the rocprofv3 --pmc counters per tabmat kernel."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    if "tmh::" not in name:
        continue
    key = name.split("tmh::")[1][:44]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in v.items():
        print(f"    {c:28s} {x / cnt[(k, c)]:.4g}")
