timeout 900 python -m pytest tests/test_cpp_adapter.py -q -m gpu -x 2>&1 | tail -4
