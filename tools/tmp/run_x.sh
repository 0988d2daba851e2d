timeout 1200 python -m pytest tests/test_align_gpu.py tests/test_align_long_gpu.py tests/test_refine.py -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()], d['real_reads'] and (round(d['real_reads']['value']/1e6,2), d['real_reads']['align_stage_ms']))"
python bench.py --read-len 10000 --reads-per-gpu 100000 --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('long', d['value'], d['ms_per_step'])"
