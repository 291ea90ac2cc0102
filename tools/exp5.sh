cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for n in 100000000 1000000000; do
  SWZ_DEBUG=1 timeout 900 python bench.py --points $n --sampler MIN_DISTANCE --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep -v amdgpu.ids | cut -c1-1800 | tee -a gpurun_out/exp5.txt
done
