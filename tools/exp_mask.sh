run() { echo "== $*"; env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-raster-subreport 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k=d['roofline']['all_kernels']
print(d['value'], d['step_ms'], 'fbi', k['fb_iteration_fused']['ms_per_step'], 'ws', k['ws_relax_sweep']['ms_per_step'], 'sor', k['vr_sor']['ms_per_step'], 'mem', d['peak_device_memory_GB'])
"; }
run A=1
run TF_FBI_TWO_PART_CHAIN=1 TF_WINDOWS_FLOOD_CUS=4
run TF_FBI_TWO_PART_CHAIN=1 TF_WINDOWS_FLOOD_CUS=8
run TF_FBI_TWO_PART_CHAIN=1 TF_WINDOWS_FLOOD_CUS=16
run TF_WINDOWS_FLOOD_CUS=8
run TF_FBI_TWO_PART_CHAIN=1
