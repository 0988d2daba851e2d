#!/bin/bash
# rocprofv3 kernel statistics of a bench run (GPU box): tools/prof_stats.sh <tag> [bench args...]
# writes gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json} and prints the top kernels by total time
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$tag
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
find $O/t -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/t
python3 - "$O/kernel_stats.csv" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:30]:
    print("%-64s calls %6s avg_us %10.1f min_us %9.1f total_ms %9.2f  %5s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
P
python3 - <<PY
import json
d=json.load(open("$O/bench.json")); print({k:d.get(k) for k in ("value","ms_per_step","value_device_resident","kernels_ms")})
PY
