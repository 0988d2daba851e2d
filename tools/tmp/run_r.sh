python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -12
