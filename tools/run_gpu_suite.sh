#!/bin/bash
# The whole `-m gpu` suite in one process (GPU box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
