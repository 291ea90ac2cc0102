# per-level MIN_DISTANCE debug lines of bench.py for the default library and every libswz_v*.so variant
cd $GRAFT_REPO_ROOT
for lib in schwarzwald_amd/lib/libswz_gpu.so schwarzwald_amd/lib/libswz_v*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  SWZ_DEBUG=1 SWZ_GPU_LIBRARY=$PWD/$lib timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 ${ARGS:---md-mode exact} 2>&1 | grep -E "${PATTERN:-sparse path}|ms_per_step" | tail -${TAIL:-3} | sed 's/.*rounds, //' | cut -c1-200
done
