run() { echo "== $*"; env "$@" python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-raster-subreport 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['step_ms'])
"; }
for cfg in "$@"; do if [ "$cfg" = "-" ]; then run A=1; else run $cfg; fi; done
