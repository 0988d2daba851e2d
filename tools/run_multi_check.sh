cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 300 python3 bench.py --no-cpu-baseline --no-host-to-host --total-reads 1250000 --pipelines 4 > /tmp/o.json 2>/tmp/o.err || { echo FAILED; tail -3 /tmp/o.err; }
python3 -c "
import json; d=json.load(open('/tmp/o.json')); r=d['real_reads']; print('1.25M p4 noreclaim run $i value %.1f M/s (%.2f ms) resident %.1f pipes %d | real %.1f in-flight %.1f' % (d['value']/1e6, d['ms_per_step'], d['value_device_resident']/1e6, d['config']['batches_in_flight'], r['value']/1e6, r['batches_in_flight_run']['value']/1e6))"
done
