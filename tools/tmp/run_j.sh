python -m pytest tests/test_align_gpu.py tests/test_place_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['real_reads']['value'])"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['value_host_to_host'])"
