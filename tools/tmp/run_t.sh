timeout 900 python -m pytest tests/test_align_gpu.py tests/test_refine.py -x -q -m gpu 2>&1 | tail -4
python tools/real_reads.py 3 2>&1 | tail -4
