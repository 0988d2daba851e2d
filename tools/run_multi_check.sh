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
run "2.5M p2" X=1 -- --total-reads 2500000
run "2.5M p3 seedpar1" PMX_SEED_PAR=1 -- --total-reads 2500000 --pipelines 3
run "2.5M p4 seedpar1" PMX_SEED_PAR=1 -- --total-reads 2500000 --pipelines 4
run "5M p2" X=1 -- --total-reads 5000000
run "5M p3 seedpar1" PMX_SEED_PAR=1 -- --total-reads 5000000 --pipelines 3
run "5M p4 seedpar1" PMX_SEED_PAR=1 -- --total-reads 5000000 --pipelines 4
run "10M p2 seedpar1" PMX_SEED_PAR=1 --
run "10M p2" X=1 --
