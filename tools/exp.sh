cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mb -o mb -- python3 $GRAFT_REPO_ROOT/bench.py --points 100000000 --batches 10 --sampler RANDOM_GRID --steps 2 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/mb_run.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/prof_mb -name "*kernel_trace*" -delete
