python -m pytest tests/test_refine.py -x -q -m gpu 2>&1 | tail -15
