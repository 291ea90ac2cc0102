# Copies the summaries of the last tools/profile_round.sh run (gpurun_out/profile/) into profiles/$1 (default r04) and derives
# traffic.json; files other tools left there (las_decode_*.txt, bin_backend_*.txt) are kept.
R=${1:-r04}
P=gpurun_out/profile
D=profiles/$R
mkdir -p /tmp/keep_$R && cp $D/las_decode_*.txt $D/bin_backend_*.txt /tmp/keep_$R/ 2>/dev/null
rm -rf $D && mkdir -p $D && cp /tmp/keep_$R/* $D/ 2>/dev/null
cp $P/bench_100M_*.json $P/bench_1B_100batches_*_tiles.json $P/bench_1B_100batches_*_uniform.json $P/bench_1B_GRID_CENTER.json \
   $P/bench_1B_JITTERED.json $P/bench_1B_RANDOM_GRID.json $P/bench_1B_min_distance*.json $P/bench_500M_*.json $P/bench_group_driver_*.json \
   $P/clustered_100M.txt $P/fullsize_pytest.txt $P/fullsize_verification_1B.log $P/group_joint_root_vs_turns.txt $P/pmc_*_by_kernel*.csv \
   $P/source_sha16.txt $D/
cp $P/stats_run.json $D/bench_under_rocprofv3.json
cp $P/stats/bench_kernel_stats.csv $D/rocprofv3_kernel_stats_bench_default.csv
cp $P/stats_gc/bench_kernel_stats.csv $D/rocprofv3_kernel_stats_GRID_CENTER.csv
cp $P/stats_prop/bench_kernel_stats.csv $D/rocprofv3_kernel_stats_property_mode.csv
cp $P/stats_mb/bench_kernel_stats.csv $D/rocprofv3_kernel_stats_1B_100batches_RANDOM_GRID.csv
python tools/make_traffic.py $D
echo "run on sources $(cat $D/source_sha16.txt), tree has $(python -c 'import bench; print(bench.library_source_sha16())')"
