#!/bin/bash
# rocprofv3 evidence for the two workloads that are not the headline: the repository's real reads (DP service, general tiers)
# and long reads (config 4, wave-per-read kernel).  Kernel stats and the SQ counters in separate runs.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03_other
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rr_stats -- python3 tools/real_reads.py > $O/real_reads.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/rr_sq -- python3 tools/real_reads.py > $O/real_reads_sq.log 2>&1
LR="bench.py --scaling weak --reads-per-gpu 8000 --read-len 10000 --steps 2 --warmup 1 --no-cpu-baseline --pipelines 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lr_stats -- python3 $LR > $O/long_reads.json 2> $O/long_reads.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/lr_sq -- python3 $LR > $O/long_reads_sq.json 2> $O/long_reads_sq.log
find $O/rr_stats -name "*kernel_stats.csv" -exec cp {} $O/real_reads_kernel_stats.csv \;
find $O/lr_stats -name "*kernel_stats.csv" -exec cp {} $O/long_reads_kernel_stats.csv \;
for d in rr_sq lr_sq; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 - "$f" "$O/pmc_sq_$d.csv" <<'PY'
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
done
rm -rf $O/rr_stats $O/rr_sq $O/lr_stats $O/lr_sq
ls -la $O; head -8 $O/real_reads_kernel_stats.csv | cut -c1-140; head -6 $O/long_reads_kernel_stats.csv | cut -c1-140; tail -3 $O/real_reads.log
