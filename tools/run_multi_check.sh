cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3 4 5 6; do
timeout 300 python3 bench.py --no-real-reads --no-cpu-baseline --no-host-to-host --total-reads 1250000 > /tmp/o.json 2>/tmp/o.err || { echo FAILED; tail -3 /tmp/o.err; }
python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('1.25M run $i value %.1f M/s (%.2f ms) resident %.1f pipes %d' % (d['value']/1e6, d['ms_per_step'], d['value_device_resident']/1e6, d['config']['batches_in_flight']))"
done
for i in 1 2; do
timeout 300 python3 bench.py --no-real-reads --no-cpu-baseline --no-host-to-host --total-reads 2500000 > /tmp/o.json 2>/tmp/o.err || { echo FAILED; tail -3 /tmp/o.err; }
python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('2.5M run $i value %.1f M/s (%.2f ms) resident %.1f pipes %d' % (d['value']/1e6, d['ms_per_step'], d['value_device_resident']/1e6, d['config']['batches_in_flight']))"
done
