python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_align_long_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
PMX_ALIGN_HOST_INDEX=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('host index', d['value'], d['ms_per_step'], d['kernels_ms'])"
