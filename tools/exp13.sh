cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
SWZ_DEBUG=1 timeout 300 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep:|rounds|metric" | cut -c1-160 | tail -9 | tee -a gpurun_out/exp13.txt
