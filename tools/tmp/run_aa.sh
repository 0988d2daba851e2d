PMX_ALIGN_VERBOSE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --no-real-reads 2>gpurun_out/aa.err | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()], d['checks']['tiers'])"
grep "wave-tier" gpurun_out/aa.err | tail -1
timeout 1200 python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -3
