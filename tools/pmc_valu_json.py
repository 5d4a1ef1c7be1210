"""Summarise a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE pass of bench.py into per-kernel VALU issue figures:

  python tools/pmc_valu_json.py <dir of the pass> <out.json> <JSON line bench.py printed in that pass>

per kernel: wave-level VALU instructions per launch (SQ_INSTS_VALU is summed over the 8 XCDs), waves per launch, busy cycles
per launch (GRBM_GUI_ACTIVE / 8) and -- with the pass's own dispatch timestamps -- the clock the chip ran at under that kernel.
`recorded_on` carries the algorithmic bytes per launch of the same run, so bench.py can turn "instructions per algorithmic
byte" into the VALU ISSUE FLOOR of any run: instructions x 4 cycles (one wave64 instruction occupies a SIMD for four cycles;
float64 instructions take eight, so this is a lower bound) / (256 CUs x 4 SIMDs x clock)."""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    if "SQ_INSTS_VALU" not in c:
        continue
    n = len(c["SQ_INSTS_VALU"])
    mean = lambda name: (sum(c[name]) / len(c[name])) if c.get(name) else None
    gui = mean("GRBM_GUI_ACTIVE")
    d = {"launches": n, "valu_wave_instructions_per_launch": mean("SQ_INSTS_VALU"), "waves_per_launch": mean("SQ_WAVES"),
         "busy_cycles_per_launch": gui / 8.0 if gui else None}
    if dur.get(k) and gui:
        ns = sum(dur[k]) / len(dur[k])
        d["clock_ghz_under_this_kernel"] = round(gui / 8.0 / ns, 3) if ns > 0 else None
    out[k.split("(")[0][:80]] = d
line = [l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1]
b = json.loads(line)
recorded = {"workload": b["config"]["workload"],
            "algorithmic_bytes_per_launch": {k: v["algorithmic_bytes_per_launch"] for k, v in b["roofline"]["all_kernels"].items()},
            "launches": {k: v["launches"] for k, v in b["roofline"]["all_kernels"].items()}}
json.dump({"unit": "wave-level VALU instructions per launch, mean over all launches of the run", "simds": 1024, "cycles_per_wave64_instruction": 4,
           "recorded_on": recorded, "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["valu_wave_instructions_per_launch"] or 0) * kv[1]["launches"])[:14]:
    print(f"{k[:50]:50s} n={v['launches']:5d} valu/launch={v['valu_wave_instructions_per_launch']:.4g} waves={v['waves_per_launch'] or 0:.4g} clock={v.get('clock_ghz_under_this_kernel')}")
