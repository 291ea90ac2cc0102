export SWZ_MD_TIME_LIMIT=20
SWZ_DEBUG=1 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep|sparse|ms_per_step" | tail -6 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['kernels_ms_per_step'])
    else:
        print(l.rstrip()[:160])"
timeout 1200 python -m pytest tests/test_cpp_group.py tests/test_sharded_gloo.py tests/test_min_distance_keys.py -q -m gpu -x 2>&1 | tail -4
