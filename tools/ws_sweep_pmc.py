"""Counter totals of the watershed sweep kernels' LARGE launches from a rocprofv3 --pmc run (window bench):
   python3 tools/ws_sweep_pmc.py <dir>"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("k_ws_sweep"):
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, d in acc.items():
    print(n, {k: "%.4g" % v for k, v in d.items()})
