cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace15
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_rounds.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/exp15.txt 2>&1
cp $(find $OUT -name "*kernel_stats.csv") $GRAFT_REPO_ROOT/gpurun_out/exp15_stats.csv
rm -rf $OUT
