#!/bin/bash
# `value` of a 1.25M-read job (one rank's share of the 8-GPU strong-scaling run) under a few settings (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python3 bench.py --no-real-reads --no-cpu-baseline --no-host-to-host "$@" > /tmp/o.json 2>/tmp/o.err || { echo "$label FAILED"; tail -3 /tmp/o.err; return; }
  python3 - "$label" <<'P'
import json,sys
d=json.load(open("/tmp/o.json")); k=d["kernels_ms"]
print("%-40s value %.1f M/s (%.2f ms/step)  one-at-a-time %.1f M/s  align %.2f seed %.2f same=%s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["value_device_resident"]/1e6, k["align stage (all tiers)"], k["seed stage (k_seed_histogram, chunked)"], d["equals_device_resident_run"]))
P
}
if [ "$1" = "scale" ]; then
run "5M p2" X=1 -- --total-reads 5000000
run "2.5M p2" X=1 -- --total-reads 2500000
run "1.25M p2" X=1 -- --total-reads 1250000
exit 0
fi
if [ "$1" = "10M" ]; then
run "10M p2" X=1 --
run "10M p3" X=1 -- --pipelines 3
run "10M p2 again" X=1 --
exit 0
fi
run "1.25M p2" X=1 -- --total-reads 1250000
run "1.25M p3" X=1 -- --total-reads 1250000 --pipelines 3
run "1.25M p4" X=1 -- --total-reads 1250000 --pipelines 4
run "1.25M p2 no_multi" PMX_ALIGN_NO_MULTI=1 -- --total-reads 1250000
run "2.5M p3" X=1 -- --total-reads 2500000 --pipelines 3
run "2.5M p2" X=1 -- --total-reads 2500000
