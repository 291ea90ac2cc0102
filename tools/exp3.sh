set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
for n in 10000000 100000000; do
  timeout 600 python bench.py --points $n --sampler MIN_DISTANCE --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | tee -a gpurun_out/exp3.jsonl
done
