#!/bin/bash
# SQ counters of the long-read kernel (config 4 at 8,000 reads); gpurun_out/prof_lr/pmc_sq.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_lr
mkdir -p $O
cd $R
LR="bench.py --scaling weak --reads-per-gpu 8000 --read-len 10000 --steps 2 --warmup 1 --no-cpu-baseline --pipelines 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $LR > $O/run_sq.json 2> $O/run_sq.err
f=$(find $O/sq -name "*counter_collection.csv" | head -1)
python3 - "$f" "$O/pmc_sq.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    if "k_align" not in r["Kernel_Name"]: continue
    key = (r["Kernel_Name"][:48], r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Mean_per_dispatch"])
for (k, c), (n, s) in sorted(agg.items()): w.writerow([k, c, n, s, s / n])
PY
rm -rf $O/sq
cat $O/pmc_sq.csv
