python -m pytest tests/test_cli.py tests/test_refine.py -x -q -m gpu 2>&1 | tail -4
python tools/refine_timing.py 2>&1 | tail -2 | cut -c1-200
