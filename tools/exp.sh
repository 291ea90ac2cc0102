for s in MIN_DISTANCE; do
timeout 900 python bench.py --points 1000000000 --batches 100 --sampler $s --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$s 1B/100 cold', d['ms_per_step'], 'ms', d['value'], 'Mpts/s')"
done
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --md-mode both 2>/dev/null | cut -c1-160
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_multibatch.py tests/test_min_distance_property.py -q -m gpu -x 2>&1 | tail -2
