"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes -- the TCC block cannot hold both) of the
same command into per-kernel HBM traffic per launch, with the gfx950 correction of MI355X_MICROARCH.md section HBM
(FETCH_SIZE reports half the bytes of wide coalesced reads; both counters are in KB).

  python tools/pmc_traffic_json.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json> [skip_first_n_launches_fraction]
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    short = k.split("(")[0][:80]
    out[short] = {"launches": max(len(f), len(w)),
                  "fetch_bytes_raw_per_launch": sum(f) / len(f) * 1024 if f else None,
                  "fetch_bytes_corrected_per_launch": sum(f) / len(f) * 2048 if f else None,
                  "write_bytes_per_launch": sum(w) / len(w) * 1024 if w else None}
# optional 4th argument: the JSON line bench.py printed in one of the counter passes -- the workload the passes ran on and the
# algorithmic bytes per launch of its dominant kernel, so that bench.py can scale the per-launch traffic to a run whose
# launches hold a different number of frame pairs (bytes per launch are proportional to the pairs in the launch)
recorded = None
if len(sys.argv) > 4:
    try:
        line = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        recorded = {"workload": b["config"]["workload"],
                    "algorithmic_bytes_per_launch": {k: v["algorithmic_bytes_per_launch"] for k, v in b["roofline"]["all_kernels"].items()},
                    "launches": {k: v["launches"] for k, v in b["roofline"]["all_kernels"].items()}}
    except (OSError, ValueError, KeyError, IndexError):
        recorded = None
json.dump({"unit": "bytes per launch, mean over all launches of the run (warm-up included)",
           "recorded_on": recorded,
           "correction": "FETCH_SIZE x 2 for 16 B / lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_corrected_per_launch"] or 0) * kv[1]["launches"])[:14]:
    print(f"{k[:50]:50s} n={v['launches']:5d} fetch_corr={v['fetch_bytes_corrected_per_launch'] or 0:.4g} write={v['write_bytes_per_launch'] or 0:.4g}")
