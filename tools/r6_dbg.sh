# round 6: timing experiments on the block kernel: SWZ_SP_BLOCK_DBG switches that break the result but show what a phase costs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for d in ${DBGS:-0 1 2 4}; do
  echo "== dbg $d"
  SWZ_SP_BLOCK_DBG=$d SWZ_GPU_LIBRARY=$PWD/schwarzwald_amd/lib/libswz_vstats.so SWZ_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact --also "" > gpurun_out/r6/dbg$d.json 2> gpurun_out/r6/dbg$d.err
  grep -E "block path|thread 0" gpurun_out/r6/dbg$d.err | cut -c1-60,200-420 | tail -4
done
