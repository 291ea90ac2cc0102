# Per-level timing of the MIN_DISTANCE headline workload (debug prints of the library, SWZ_DEBUG=1) after the
# GPU parity tests.  Extra environment (SWZ_MD_* scheduling overrides) is passed through.  Usage via gpurun:
#   gpurun --timeout 900 -- 'bash tools/md_levels.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
SWZ_DEBUG=1 timeout 300 python bench.py --points ${POINTS:-1000000000} --sampler MIN_DISTANCE --steps 2 --warmup 1 --cpu-sample 0 2>&1 \
  | grep -E "sweep:|rounds|sparse path|metric" | tail -9 | cut -c1-170 | tee gpurun_out/md_levels.txt
