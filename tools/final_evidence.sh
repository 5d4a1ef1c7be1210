# End-of-round refresh after the counter passes are in profiles/: smoke, the default bench line, the rocprofv3 kernel summary of the
# same command (no --pmc).  Usage: bash tools/final_evidence.sh <tag>
tag=${1:-r5f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.txt 2>&1; tail -1 gpurun_out/${tag}_smoke.txt
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.json; echo
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err
python tools/shorten_kernel_stats.py $(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1) > gpurun_out/${tag}_kernel_stats.csv
head -6 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/${tag}_prof
