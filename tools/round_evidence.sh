cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_final_smoke.txt 2>&1; tail -1 gpurun_out/r2_final_smoke.txt
python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err; tail -c 600 gpurun_out/r2_final_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_final_prof -- python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r2_final_bench_under_rocprof.json 2> gpurun_out/r2_final_prof.err
python tools/shorten_kernel_stats.py $(ls gpurun_out/r2_final_prof/*/*kernel_stats.csv | head -1) > gpurun_out/r2_final_kernel_stats.csv
head -12 gpurun_out/r2_final_kernel_stats.csv | cut -c1-160
