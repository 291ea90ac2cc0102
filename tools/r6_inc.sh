# round 6: MIN_DISTANCE on a batch merged with files -- only what the batch can change (sb_incremental) against the whole level,
# same box: tests first, then 1 B points in 100 uniform / tile batches with and without, then a debug log of a short run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_multibatch.py tests/test_min_distance_blocks.py -x -q -m gpu --durations=8 2>&1 | tail -16
for order in ${INC_ORDERS:-uniform tiles}; do
  for inc in ${INC_VARIANTS:-default 0}; do
    f=gpurun_out/r6/inc_${order}_${inc}.json
    if [ "$inc" = default ]; then unset SWZ_SP_INCREMENTAL; else export SWZ_SP_INCREMENTAL=$inc; fi
    timeout 900 python bench.py --batches 100 --strategy FAST --batch-order $order --md-mode exact --steps 2 --warmup 1 --cpu-sample 0 > $f 2> ${f%.json}.err
    python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "first_data_set_ms", d.get("first_data_set_ms"),
      "V", d.get("visit_factor"), {k: round(v) for k, v in d["kernels_ms_per_step"].items()})
PY
  done
done
unset SWZ_SP_INCREMENTAL
SWZ_DEBUG=1 timeout 600 python bench.py --batches 100 --strategy FAST --batch-order uniform --md-mode exact --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/r6/inc_dbg.json 2> gpurun_out/r6/inc_dbg.err
grep -c "can change" gpurun_out/r6/inc_dbg.err
grep "MIN_DISTANCE level" gpurun_out/r6/inc_dbg.err | tail -40
