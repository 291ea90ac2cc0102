cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_mb; rm -rf $OUT; mkdir -p $OUT
B() { tag=$1; shift; timeout 1500 python bench.py "$@" 2>> $OUT/bench.err | grep '^{' | tail -1 > $OUT/bench_$tag.json; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$tag.json | head -1)"; }
for o in tiles uniform; do
  B 1B_100batches_MIN_DISTANCE_FAST_$o --points 1000000000 --batches 100 --batch-order $o --strategy FAST --steps 2 --warmup 1 --cpu-sample 0 --md-mode exact
  B 1B_100batches_MIN_DISTANCE_FAST_${o}_property --points 1000000000 --batches 100 --batch-order $o --strategy FAST --steps 2 --warmup 1 --cpu-sample 0 --md-mode property
  B 1B_100batches_MIN_DISTANCE_$o --points 1000000000 --batches 100 --batch-order $o --steps 1 --warmup 1 --cpu-sample 0 --md-mode exact
  B 1B_100batches_RANDOM_GRID_$o --points 1000000000 --batches 100 --batch-order $o --sampler RANDOM_GRID --steps 2 --warmup 1 --cpu-sample 0
done
