# Round evidence on one MI355X box (run through gpurun): smoke, the default bench line, the rocprofv3 kernel summary of the
# same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains besides the kernel trace).
# The counter passes use a 62-frame stack (61 pairs = 3 Farneback batches of 20 / 21 pairs, the batch size of the full
# 144-frame run): rocprofiler's counter collection does not survive ~10^4 dispatches per process on this stack
# (profiles/README.md), a full config-F step has ~10.5 k.  Usage: bash tools/round_evidence.sh <tag>
tag=${1:-r6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; tail -1 gpurun_out/${tag}_smoke.txt
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 400 gpurun_out/${tag}_bench.json; echo
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
python tools/shorten_kernel_stats.py $(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) > gpurun_out/${tag}_kernel_stats.csv
head -8 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/${tag}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python3 bench.py --frames 62 --n-windows 5 --steps 1 --warmup 0 --no-cpu-baseline --no-raster-subreport > gpurun_out/${tag}_pmc_$c.json 2> gpurun_out/${tag}_pmc_$c.err || echo "pmc pass $c failed"
done
python tools/pmc_traffic_json.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_traffic_bench.json gpurun_out/${tag}_pmc_FETCH_SIZE.json | head -8
rm -rf gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE
# VALU issue floor per kernel (round 4): wave-level VALU instructions, waves and busy cycles of the same reduced step
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_VALU -- python3 bench.py --frames 62 --n-windows 5 --steps 1 --warmup 0 --no-cpu-baseline --no-raster-subreport > gpurun_out/${tag}_pmc_VALU.json 2> gpurun_out/${tag}_pmc_VALU.err || echo "pmc pass VALU failed"
python tools/pmc_valu_json.py gpurun_out/${tag}_pmc_VALU gpurun_out/${tag}_pmc_valu_bench.json gpurun_out/${tag}_pmc_VALU.json | head -10
rm -rf gpurun_out/${tag}_pmc_VALU
