python -m pytest tests/test_place_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()])"; }
run base
PMX_SEED_NO_HINT=1 run nohint
PMX_SEED_SORT_BITS=24 PMX_ALIGN_SORT_BITS=24 run bits24
PMX_SEED_SORT_BITS=16 PMX_ALIGN_SORT_BITS=16 run bits16
PMX_SEED_SORT_BITS=16 PMX_ALIGN_SORT_BITS=8 run bits16_8
PMX_ALIGN_NO_PAIR_SORT=1 run nopairsort
run base
