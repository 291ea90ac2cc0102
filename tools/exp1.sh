set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for s in RANDOM_GRID GRID_CENTER JITTERED MIN_DISTANCE; do
  timeout 300 python bench.py --points 10000000 --sampler $s --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | tee -a gpurun_out/exp1.jsonl
done
for s in RANDOM_GRID GRID_CENTER MIN_DISTANCE; do
  timeout 600 python bench.py --points 100000000 --sampler $s --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | tee -a gpurun_out/exp1.jsonl
done
