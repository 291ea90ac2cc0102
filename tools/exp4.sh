cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SWZ_DEBUG=1 timeout 600 python bench.py --points 100000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -v amdgpu.ids | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o p4 -- python3 $GRAFT_REPO_ROOT/bench.py --points 100000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT/gpurun_out/prof4 && ls -R | head; find . -name "*kernel_stats*" | head -1 | xargs head -20 | cut -c1-200
find . -name "*kernel_trace*" -delete
