cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python3 bench.py --no-real-reads --no-cpu-baseline --no-host-to-host "$@" > /tmp/o.json 2>/tmp/o.err || { echo "$label FAILED"; tail -3 /tmp/o.err; return; }
  python3 - "$label" <<'P'
import json,sys
d=json.load(open("/tmp/o.json")); k=d["kernels_ms"]
print("%-40s value %.1f M/s (%.2f ms/step)  one-at-a-time %.1f M/s  align %.2f seed %.2f pipes %d" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["value_device_resident"]/1e6, k["align stage (all tiers)"], k["seed stage (k_seed_histogram, chunked)"], d["config"]["batches_in_flight"]))
P
}
run "1.25M default" X=1 -- --total-reads 1250000
run "1.25M p4 seedpar1" PMX_SEED_PAR=1 -- --total-reads 1250000 --pipelines 4
run "1.25M p3" X=1 -- --total-reads 1250000 --pipelines 3
run "1.25M default again" X=1 -- --total-reads 1250000
run "1.25M default steps40" X=1 -- --total-reads 1250000 --steps 40 --warmup 8
