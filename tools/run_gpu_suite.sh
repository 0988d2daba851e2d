#!/bin/bash
# The whole `-m gpu` suite in one process (GPU box); the full log goes to gpurun_out/gpu_suite.log.
# tools/run_gpu_suite.sh [pytest args...]   (default: tests)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2700 python -m pytest ${@:-tests} -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1
tail -15 gpurun_out/gpu_suite.log
