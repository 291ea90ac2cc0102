# rocprofv3 kernel summary of one bench.py command: ARGS in the environment, output gpurun_out/r5/prof_<TAG>.csv
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5
mkdir -p $OUT
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_$TAG.json 2> $OUT/prof_$TAG.err
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/prof_$TAG.csv
head -25 $OUT/prof_$TAG.csv | cut -c1-160
