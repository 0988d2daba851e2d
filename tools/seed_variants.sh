#!/bin/bash
# seeding-stage time of the default workload under a few settings (GPU box); every run under a timeout
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "X=1" "PMX_SEED_PAR=1" "PMX_SEED_NO_COLLAPSE=1"; do
  env $v timeout 200 python3 bench.py --steps 6 --warmup 2 --no-real-reads --no-cpu-baseline --pipelines 1 "$@" > /tmp/o.json 2>/tmp/o.err || { echo "$v FAILED rc=$?"; tail -3 /tmp/o.err; continue; }
  python3 - "$v" <<'P'
import json,sys
d=json.load(open("/tmp/o.json")); k=d["kernels_ms"]
print("%-45s seed %.2f align %.2f resident %.1f M/s (%.2f ms) same=%s" % (sys.argv[1] or "default", k["seed stage (k_seed_histogram, chunked)"], k["align stage (all tiers)"], d["value_device_resident"]/1e6, d["device_resident"]["ms_per_step"], d["equals_device_resident_run"]))
P
done
