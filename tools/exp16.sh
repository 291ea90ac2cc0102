cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SWZ_DEBUG=1 timeout 300 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "sweep:|rounds|metric" | grep -v sparse | cut -c1-250 | tee gpurun_out/exp16.txt
