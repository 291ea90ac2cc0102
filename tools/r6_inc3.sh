# round 6: tuning of the subset path on 1 B points in 100 uniform FAST batches (same box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
run() {
  name=$1; shift
  f=gpurun_out/r6/inc3_$name.json
  env "$@" timeout 900 python bench.py --batches 100 --strategy FAST --batch-order uniform --md-mode exact --steps 2 --warmup 1 --cpu-sample 0 > $f 2> ${f%.json}.err
  python - "$f" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms_per_step", d["ms_per_step"], {k: round(v) for k, v in d["kernels_ms_per_step"].items()})
PY
}
run default A=1
run spread4 SWZ_SP_INCREMENTAL_SPREAD=4
run min64 SWZ_SP_BLOCK_MIN=64
run min256 SWZ_SP_BLOCK_MIN=256
run default_again A=1
