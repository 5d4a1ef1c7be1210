"""k_fb_iter launches of a rocprofv3 --pmc run grouped by grid size: mean counter values per launch.
   python3 tools/fb_levels_pmc.py <dir>"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fb_iter" not in r["Kernel_Name"]:
            continue
        g = r.get("Grid_Size") or (r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?") + "x" + r.get("Grid_Size_Z", "?"))
        acc[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, d in sorted(acc.items(), key=lambda kv: -max(len(v) and sum(v) for v in kv[1].values())):
    print("grid", g, {k: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for k, v in d.items()})
