set -x
python -m pytest tests/test_place_gpu.py -x -q -m gpu -k "e2e or rsv" 2>&1 | tail -3
for L in 64 16 8 4; do
  PMX_ALIGN_TPP_MIN_LANES=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('L=$L', d['value'], d['ms_per_step'], d['kernels_ms'])"
done
PMX_ALIGN_TPP_MIN_LANES=8 python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -3
