run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --no-real-reads 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()], d['checks']['tiers']['wave_tier_items'])"; }
export PMX_ALIGN_NO_COMPACT_WIDE=1 PMX_ALIGN_BAIL_TPP_MIN=100000
run lds24
PMX_ALIGN_LDS_KB=16 run lds16
PMX_ALIGN_LDS_KB=12 run lds12
PMX_ALIGN_LDS_KB=32 run lds32
PMX_ALIGN_LDS_KB=20 run lds20
