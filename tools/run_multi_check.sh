cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_zz_compact_forms_gpu.py tests/test_align_gpu.py -x -q 2>&1 | tail -3
run() { label=$1; shift; env "$@" timeout 600 python3 bench.py --no-cpu-baseline --no-host-to-host --no-real-reads > gpurun_out/m_$label.json 2> gpurun_out/m_$label.err || tail -3 gpurun_out/m_$label.err
python3 - $label <<'P'
import json,sys
d=json.load(open("gpurun_out/m_%s.json"%sys.argv[1])); k=d["kernels_ms"]
print("%-10s value %.1f M/s  %.2f ms  resident %.1f  align %.2f chain %.2f cseeds %.2f seed %.2f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["value_device_resident"]/1e6, k["align stage (all tiers)"], k["dominant align kernel"], k["compact tier, sketch + probes (k_compact_seeds16)"], k["seed stage (k_seed_histogram, chunked)"]))
P
}
run staged X=1
run rows PMX_LIB_PATH=$PWD/panmap_amd/libpanmap_amd_rows.so
run staged2 X=1
run rows2 PMX_LIB_PATH=$PWD/panmap_amd/libpanmap_amd_rows.so
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_ho
rm -rf $O && mkdir -p $O
cd $R
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --pipelines 1"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $ARGS > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $ARGS > $O/write.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys,re
O=sys.argv[1]
def per(d,counter,pat):
    vals={}
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==counter and re.search(pat,r["Kernel_Name"]):
                vals[int(r["Dispatch_Id"])]=vals.get(int(r["Dispatch_Id"]),0.0)+float(r["Counter_Value"])
    return list(vals.values())
for name,pat in (("k_compact_seeds16","k_compact_seeds16"),("k_align_compact16",r"k_align_compact16\("),("multi","k_align_compact16_multi")):
    f=per(O+"/fetch","FETCH_SIZE",pat); w=per(O+"/write","WRITE_SIZE",pat)
    if f and w: print(name, "launches", len(f), "hbm GB per launch %.3f (fetch x2 %.3f + write %.3f)" % ((2*sum(f)/len(f)+sum(w)/len(w))*1024/1e9, 2*sum(f)/len(f)*1024/1e9, sum(w)/len(w)*1024/1e9))
PY
rm -rf $O/fetch $O/write
