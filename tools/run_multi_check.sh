cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { label=$1; shift; env "$@" timeout 600 python3 bench.py --no-cpu-baseline --no-host-to-host --no-real-reads > gpurun_out/m_$label.json 2> gpurun_out/m_$label.err || tail -3 gpurun_out/m_$label.err
python3 - $label <<'P'
import json,sys
d=json.load(open("gpurun_out/m_%s.json"%sys.argv[1]))
print("%-12s value %.1f M/s  %.2f ms  resident %.1f  align %.2f  steps %d warmup %d" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["value_device_resident"]/1e6, d["kernels_ms"]["align stage (all tiers)"], d["steps"], d["warmup"]))
P
}
run default X=1
run nomulti PMX_ALIGN_NO_MULTI=1
run default2 X=1
