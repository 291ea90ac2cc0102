# round 6: the reference's default operating point (MIN_DISTANCE, FAST, 1 B points in 100 batches) -- tiles and uniform batches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for order in tiles uniform; do
  for mode in exact ${MB_MODES:-}; do
    f=gpurun_out/r6/mb_${order}_${mode}.json
    timeout 900 python bench.py --batches 100 --strategy FAST --batch-order $order --md-mode $mode --steps 2 --warmup 1 --cpu-sample 0 > $f 2> ${f%.json}.err
    python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "first_data_set_ms", d.get("first_data_set_ms"), "extra warmups", d.get("extra_warmup_data_sets"),
      "grew", d.get("workspace_grew_in_timed_steps"), "V", d.get("visit_factor"), {k: round(v) for k, v in d["kernels_ms_per_step"].items()})
PY
  done
done
