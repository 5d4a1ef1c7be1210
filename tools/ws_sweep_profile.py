"""Launch durations of the watershed sweep kernels from a rocprofv3 kernel trace, largest first, per kernel:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --config window --frames 16 --steps 1 --warmup 1 --no-cpu-baseline
   python3 tools/ws_sweep_profile.py <dir>"""
import csv, glob, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("k_ws_") or "rocprim" in n:
            d[n[:40]].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
for n, v in sorted(d.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    v.sort(reverse=True)
    tot = sum(x[0] for x in v)
    print("%-40s launches %5d total %8.1f us; ten largest (us @ workgroups): %s" % (n, len(v), tot, " ".join("%.0f@%d" % x for x in v[:10])))
