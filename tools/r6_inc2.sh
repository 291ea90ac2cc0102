# round 6: the incremental subset path -- tests, the two 100-batch runs, kernel statistics of the uniform one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_multibatch.py tests/test_min_distance_blocks.py -x -q -m gpu 2>&1 | tail -3
for order in uniform tiles; do
  f=gpurun_out/r6/inc_${order}_default.json
  timeout 900 python bench.py --batches 100 --strategy FAST --batch-order $order --md-mode exact --steps 2 --warmup 1 --cpu-sample 0 > $f 2> ${f%.json}.err
  python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "first_data_set_ms", d.get("first_data_set_ms"),
      "V", d.get("visit_factor"), {k: round(v) for k, v in d["kernels_ms_per_step"].items()})
PY
done
bash tools/r6_inc_prof.sh 2>&1 | awk -F, '{print $1, $2, $3, $4}' | cut -c1-160
