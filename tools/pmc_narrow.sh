#!/bin/bash
# Runs tools/pmc_narrow.py under rocprofv3 --pmc <counter> for a list of configurations, one process each, keeping stderr
# and the profiler's output directory of every run.  Stops at the first run that had to be killed (a hang), goes on
# after an ordinary non-zero exit (the segmentation fault under investigation).
#   tools/pmc_narrow.sh <counter> <outdir> "<args of case 1>" "<args of case 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
counter=$1; out=$2; shift 2
mkdir -p "$out"
i=0
for args in "$@"; do
  i=$((i+1))
  tag=$(echo "case${i}_$args" | tr ' -' '__' | tr -s '_')
  echo "=== $tag" | tee -a "$out/summary.txt"
  timeout -k 10 240 rocprofv3 --pmc $counter --output-format csv -d "$out/$tag" -- python3 -X faulthandler tools/pmc_narrow.py $args > "$out/$tag.stdout" 2> "$out/$tag.stderr"
  rc=$?
  echo "rc=$rc $(tail -n 1 "$out/$tag.stdout")" | tee -a "$out/summary.txt"
  tail -n 25 "$out/$tag.stderr" | grep -v "^$" | tail -n 12 >> "$out/summary.txt"
  ls "$out/$tag" 2>/dev/null | head -3 >> "$out/summary.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping" | tee -a "$out/summary.txt"; exit 1; fi
done
exit 0
