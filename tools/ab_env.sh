#!/bin/bash
# Same-box A/B of environment switches on one build: tools/ab_env.sh "" "PMX_ALIGN_COMPACT_FUSED=1" ...
# -> alternating bench.py runs (10M reads): host->host reads/s, ms per step, the align kernels' times
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do
  env $v timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('[$v]', round(d['value']/1e6,2), 'M reads/s', round(d['ms_per_step'],2), 'ms; resident', round(d['device_resident']['ms_per_step'],2), 'ms;', {n: round(v,2) for n,v in k.items() if isinstance(v,(int,float))}, 'real', round(d['real_reads']['value']/1e6,2), round(d['real_reads']['align_stage_ms'],2))"
done; done
