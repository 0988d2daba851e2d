cd ${GRAFT_REPO_ROOT:-/root/repo}
PMX_ALIGN_PROF=1 timeout 600 python3 bench.py --no-cpu-baseline --no-host-to-host --no-real-reads --steps 2 --warmup 1 --pipelines 1 > gpurun_out/m_prof.json 2> gpurun_out/m_prof.err
grep "compact tier" gpurun_out/m_prof.err | tail -1
timeout 600 python3 bench.py --no-host-to-host --no-real-reads --cpu-sample 400000 > gpurun_out/m_b.json 2> gpurun_out/m_b.err || tail -3 gpurun_out/m_b.err
python3 - <<'P'
import json
d=json.load(open("gpurun_out/m_b.json"))
print("value %.1f M/s  %.2f ms  resident %.1f" % (d["value"]/1e6, d["ms_per_step"], d["value_device_resident"]/1e6))
print(json.dumps(d["kernels_ms"]))
print(json.dumps(d["checks"]["oracle"]))
P
