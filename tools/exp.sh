timeout 1200 python -m pytest tests/test_sharded_gloo.py -q -m gpu -x 2>&1 | tail -12
