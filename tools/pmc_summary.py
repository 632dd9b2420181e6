"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, sum of each counter."""
import csv, sys, collections
path = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"][:60]
        if filt and filt not in r["Kernel_Name"]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {v:18.0f}  (n={cnt[(k, c)]})")
