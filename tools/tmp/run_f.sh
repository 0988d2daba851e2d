cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_sq
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host > $O/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host > $O/sq2.log 2>&1
for d in sq sq2; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 - "$f" "$O/pmc_$d.csv" <<'PY'
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
rm -rf $O/sq $O/sq2
