# round 6: kernel statistics of 1 B points in 100 uniform FAST batches with the incremental subset path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6/prof_inc -o inc -- python3 $GRAFT_REPO_ROOT/bench.py --batches ${NB:-100} --strategy FAST --batch-order ${ORDER:-uniform} --md-mode exact --steps 1 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r6/prof_inc.json 2> $GRAFT_REPO_ROOT/gpurun_out/r6/prof_inc.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r6/prof_inc -name '*kernel_stats.csv' | head -1)
head -30 "$f" | cut -c1-200
find gpurun_out/r6/prof_inc -name '*.csv' ! -name '*kernel_stats.csv' -delete
find gpurun_out/r6/prof_inc -name "*kernel_trace*" -delete
