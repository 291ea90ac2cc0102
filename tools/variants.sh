# Times one sampler of bench.py with each library variant in schwarzwald_amd/lib/libswz_v*.so (experiments only) and with
# the default library.  SAMPLER / ARGS from the environment.
cd $GRAFT_REPO_ROOT
for lib in schwarzwald_amd/lib/libswz_gpu.so schwarzwald_amd/lib/libswz_v*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  SWZ_GPU_LIBRARY=$PWD/$lib timeout 300 python bench.py --sampler ${SAMPLER:-RANDOM_GRID} --steps 3 --warmup 1 --cpu-sample 0 ${ARGS:-} 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_ms_per_step".*' | cut -c1-400
done
