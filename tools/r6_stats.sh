# round 6: phase times of the block kernel (variant library built with -DSWZ_SB_STATS)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
SWZ_GPU_LIBRARY=$PWD/schwarzwald_amd/lib/libswz_vstats.so SWZ_DEBUG=1 timeout 900 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --md-mode exact --also "" $BENCH_ARGS > gpurun_out/r6/stats.json 2> gpurun_out/r6/stats.err
grep -E "block path|thread 0" gpurun_out/r6/stats.err | tail -4
