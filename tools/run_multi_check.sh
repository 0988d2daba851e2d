cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_dist_c_gpu.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -4
