for cfg in "16 3" "48 3" "150 1" "50 3" "75 2" "25 4"; do
  set -- $cfg
  PMX_SEED_CHUNK_MB=$1 PMX_SEED_PAR=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('chunk_mb=$1 par=$2', d['value'], d['ms_per_step'], d['kernels_ms'])"
done
