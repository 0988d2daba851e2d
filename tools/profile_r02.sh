#!/bin/bash
# rocprofv3 passes behind profiles/r02 (run on a GPU box from the repo root; outputs under gpurun_out/prof_r02)
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host --no-overlap > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --no-overlap > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --no-overlap > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --no-overlap > $O/sq.log 2>&1
python3 profiles/r02/make_pmc_traffic.py $O/fetch $O/write 1000000 150 $O/pmc_traffic.json > $O/traffic.log 2>&1
# keep the summaries small enough to travel back
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for d in fetch write sq; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 - "$f" "$O/pmc_$d.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_align", "k_seed_histogram", "k_score"))]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value"], extrasaction="ignore")
w.writeheader()
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"][:60]
    w.writerow(r)
PY
done
rm -rf $O/stats $O/fetch $O/write $O/sq
ls -la $O
