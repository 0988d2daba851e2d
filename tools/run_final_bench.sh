#!/bin/bash
# The bench lines kept under profiles/rNN/: the driver's command and a rank's share of the 8-GPU strong-scaling job (GPU box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default_r04.json 2> gpurun_out/bench_default_r04.err || tail -5 gpurun_out/bench_default_r04.err
timeout 600 python3 bench.py --gpus 1 --total-reads 1250000 > gpurun_out/bench_1250k_r04.json 2> gpurun_out/bench_1250k_r04.err || tail -5 gpurun_out/bench_1250k_r04.err
python3 - <<'P'
import json
for f in ("bench_default_r04","bench_1250k_r04"):
    d=json.load(open("gpurun_out/%s.json"%f))
    print(f, "value %.1f M/s %.2f ms; h2h %.1f; resident %.1f; pipes %d" % (d["value"]/1e6, d["ms_per_step"], (d.get("value_host_to_host") or 0)/1e6, d["value_device_resident"]/1e6, d["config"]["batches_in_flight"]))
    print("  kernels", json.dumps(d["kernels_ms"]))
    print("  real", json.dumps({k:v for k,v in d["real_reads"].items() if k in ("value","ms_per_step","align_stage_ms","batches_in_flight_run")}))
    print("  oracle", d["checks"]["oracle"]["records_equal"], d["checks"]["oracle"]["cigars_equal"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
    print("  roofline", json.dumps(d["roofline"]))
    print("  dp_service", d["dp_service"]["gcups"])
P
