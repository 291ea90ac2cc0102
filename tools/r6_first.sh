# round 6: first look at the block path (swz_mdblock.hip): parity on the MIN_DISTANCE suites, then per-level timings at 1 B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_min_distance_keys.py -x -q -m gpu -k "min_distance or MIN_DISTANCE or sparse or tile_accurate or fast" 2>&1 | tail -15
SWZ_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --md-mode exact --also "" > gpurun_out/r6/first_exact.json 2> gpurun_out/r6/first_exact.err
grep -E "block path|sparse path" gpurun_out/r6/first_exact.err | tail -8
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r6/first_exact.json | head -1
grep -o '"kernels_ms_per_step": {[^}]*}' gpurun_out/r6/first_exact.json | head -1
