cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for s in RANDOM_GRID GRID_CENTER; do
timeout 900 python bench.py --sampler $s --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['sampler'], d['value'], 'Mpts/s', d['ms_per_step'], 'ms', d['kernels_ms_per_step'])"
done
timeout 900 python bench.py --points 100000000 --sampler GRID_CENTER --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('100M', d['config']['sampler'], d['value'], 'Mpts/s', d['ms_per_step'], 'ms', d['kernels_ms_per_step'])"
