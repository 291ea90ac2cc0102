timeout 900 python -m pytest tests/test_multibatch.py tests/test_min_distance_property.py -q -m gpu 2>&1 | tail -3
for cfg in "100000000 10" "500000000 5"; do
    set -- $cfg
    timeout 900 python bench.py --points $1 --batches $2 --sampler MIN_DISTANCE --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MIN_DISTANCE', $1, 'points in', $2, 'batches:', d['ms_per_step'], 'ms', d['value'], 'Mpts/s')"
done
