python -m pytest tests/test_align_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],3), [round(x,3) for x in d['kernels_ms'].values()], d['checks']['tiers'], d['real_reads'] and (round(d['real_reads']['value']/1e6,2), d['real_reads']['align_stage_ms']))"; }
run base
PMX_ALIGN_BAIL_TPP_MIN=100000 run skip_tpp --no-real-reads
PMX_ALIGN_NO_COMPACT_WIDE=1 run nowide --no-real-reads
