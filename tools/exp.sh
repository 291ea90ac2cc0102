timeout 900 python -m pytest tests/test_multibatch.py -q -m gpu -x -k "spill" 2>&1 | tail -8
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
