#!/bin/bash
# Kernel statistics of the default bench workload (rocprofv3 --kernel-trace --stats); the summary goes to profiles/r03/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v1}
O=$R/gpurun_out/prof_r03_$TAG
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-real-reads > $O/bench.json 2> $O/bench.err
find $O/t -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/t
head -25 $O/kernel_stats.csv | cut -c 1-200
python3 - <<PY
import json
d=json.load(open("$O/bench.json")); print({k:d[k] for k in ("value","ms_per_step","value_device_resident","kernels_ms")})
PY
