"""Per-pyramid-level launch time of the fused Farnebaeck iteration kernel from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace -d gpurun_out/fbl -o t -- python3 tools/fb_only.py 16
   python3 tools/fb_levels.py gpurun_out/fbl
groups the k_fb_iter launches by grid size (one size per level) and prints count / average / total."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel trace under " + d)
rows = defaultdict(list)
other = defaultdict(float)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "k_fb_iter" in name:
            rows[(int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))].append(dur)
        else:
            other[name.split("(")[0][:60]] += dur
tot = sum(sum(v) for v in rows.values())
for k, v in sorted(rows.items(), key=lambda kv: -kv[0][0]):
    print("grid %s: launches %d avg %.1f us total %.2f ms (%.1f %%)" % (k, len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot))
print("k_fb_iter total %.2f ms" % (tot / 1e3))
for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-60s %.2f ms" % (k, v / 1e3))
