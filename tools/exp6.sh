cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SWZ_DEBUG=1 timeout 900 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep "sweep:" 
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -o p6 -- python3 $GRAFT_REPO_ROOT/bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/prof6 && find . -name "*kernel_trace*" -delete; ls -R | head; find . -name "*kernel_stats*" | head -1 | xargs head -30 | cut -c1-220
