# Times one sampler of bench.py with each library variant in schwarzwald_amd/lib/libswz_v*.so (experiments only).
cd $GRAFT_REPO_ROOT
for lib in schwarzwald_amd/lib/libswz_v*.so; do
  echo "== $lib"
  SWZ_GPU_LIBRARY=$PWD/$lib timeout 200 python bench.py --sampler ${SAMPLER:-GRID_CENTER} --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_ms_per_step".*'
done
