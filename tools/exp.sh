export SWZ_MD_TIME_LIMIT=20
timeout 900 python -m pytest tests/test_min_distance_keys.py tests/test_cpp_group.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -5
timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "ms_per_step" | tail -4 | cut -c1-250
