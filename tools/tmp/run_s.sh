python -m pytest tests/test_place_gpu.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()])"; }
for b in 1 2 4 8 16; do PMX_SEED_BATCHES=$b run batches$b; done
