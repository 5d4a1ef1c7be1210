# FETCH_SIZE / WRITE_SIZE of k_binary_morph16 alone (tools/morph_ab.py child: erosions of a 16 x 5424^2 volume, 3 x 3 x 3 and in-plane)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/morph_pmc_$c -- python3 tools/morph_ab.py child > gpurun_out/morph_pmc_$c.log 2>&1
  python3 - <<PY
import csv, glob, collections
tot=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob("gpurun_out/morph_pmc_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "binary_morph16" in r["Kernel_Name"] and r["Counter_Name"]=="$c":
            tot[r["Kernel_Name"][:20]][0]+=1; tot[r["Kernel_Name"][:20]][1]+=float(r["Counter_Value"])
for k,(n,v) in tot.items(): print("$c", k, "launches", n, "mean per launch", v/n)
PY
  rm -rf gpurun_out/morph_pmc_$c
done
