"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} launches {len(v):4d}  mean {sum(v) / len(v):.4g}")
