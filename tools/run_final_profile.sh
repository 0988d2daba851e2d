cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/run_final_bench.sh
bash tools/run_gpu_suite.sh
