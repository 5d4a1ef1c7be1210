# A/B of bench.py ARGUMENTS on the default step: bash tools/exp_args.sh "--chain-depth 2" "--inflight 8" ...   ("-" = defaults)
run() { echo "== $*"; python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-raster-subreport "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k=d['roofline']['all_kernels']; w=d['watershed']
print(d['value'], d['step_ms'], 'ws', k['ws_relax_sweep']['ms_per_step'], 'depth used', w['chain_depth_used_min'], w['chain_depth_used'], 'sweeps', w['sweeps_per_phase_mean_per_flood'], 'root phases', w.get('reference_order',{}).get('root_phases_per_flood_mean'), 'exact', w['labels_bit_exact_with_the_reference'])
"; }
for cfg in "$@"; do if [ "$cfg" = "-" ]; then run; else run $cfg; fi; done
