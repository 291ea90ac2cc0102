timeout 600 python -m pytest tests/test_sharded_gloo.py -q -m gpu -x -k "empty_shards and joint" 2>&1 | tail -5
