#!/bin/bash
# timeout 600 rocprofv3 passes behind profiles/r04 (run on a GPU box from the repo root; outputs under gpurun_out/prof_r04_pmc).
# The default workload of bench.py (10M reads), device-resident leg only; every PMC group in its own run.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04_pmc
rm -rf $O && mkdir -p $O
cd $R
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --pipelines 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host > $O/stats_bench.json 2> $O/stats.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $ARGS > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $ARGS > $O/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- python3 $ARGS > $O/sq.log 2>&1
python3 profiles/r04/make_pmc_traffic.py $O/fetch $O/write $O/sq 10000000 150 $O/pmc_traffic.json > $O/traffic.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for d in fetch write sq; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 - "$f" "$O/pmc_$d.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_align", "k_compact_seeds", "k_seed_histogram", "k_collapse_reads", "k_score", "k_pack_reads", "k_table_compact"))]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value"], extrasaction="ignore")
w.writeheader()
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"][:60]
    w.writerow(r)
PY
done
rm -rf $O/stats $O/fetch $O/write $O/sq
ls -la $O; tail -5 $O/traffic.log
