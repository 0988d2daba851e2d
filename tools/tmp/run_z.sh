run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --no-real-reads 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()])"; }
run dynamic
PMX_ALIGN_RESIDENT_GRID=1 run resident
run dynamic
