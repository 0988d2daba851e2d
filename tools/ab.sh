#!/bin/bash
# Same-box A/B of two builds of libpanmap_amd.so (GPU boxes differ by ~5 %): copy the builds to build_variants/ and run
#   tools/ab.sh base.so new.so      -> alternating bench.py runs, reads/s, ms/step and the align kernels' times
cd /root/repo
for r in 1 2; do for v in "$@"; do
  PMX_LIB_PATH=/root/repo/build_variants/$v timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],2), 'tpp0', round(k['k_align_reads_tpp round 0'],2), 'align', round(k['align stage (all tiers)'],2))"
done; done
