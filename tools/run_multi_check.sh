cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests/test_zz_compact_forms_gpu.py tests/test_align_gpu.py -x -q 2>&1 | tail -6
