# A/B of development switches on the default bench step: bash tools/exp_ab.sh "VAR=1" "OTHER=2 X=3" ...   ("-" = no switch)
run() { echo "== $*"; env "$@" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-raster-subreport 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
k=d['roofline']['all_kernels']
print(d['value'], d['step_ms'], ' '.join('%s %.0f' % (n[:10], k[n]['ms_per_step']) for n in k), 'mem', d['peak_device_memory_GB'])
"; }
for cfg in "$@"; do if [ "$cfg" = "-" ]; then run A=1; else run $cfg; fi; done
