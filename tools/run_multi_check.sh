cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python3 bench.py --gpus 1 --total-reads 1250000 > gpurun_out/bench_1250k_r04.json 2> gpurun_out/bench_1250k_r04.err || tail -5 gpurun_out/bench_1250k_r04.err
python3 -c "
import json; d=json.load(open('gpurun_out/bench_1250k_r04.json')); r=d['real_reads']; print('1.25M value %.1f M/s (%.2f ms) h2h %.1f resident %.1f pipes %d | real %.1f in-flight %.1f | oracle %s' % (d['value']/1e6, d['ms_per_step'], d['value_host_to_host']/1e6, d['value_device_resident']/1e6, d['config']['batches_in_flight'], r['value']/1e6, r['batches_in_flight_run']['value']/1e6, d['checks']['oracle']['records_equal']))"
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
