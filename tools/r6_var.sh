# round 6: the block kernel's level times with the default library and every variant library (schwarzwald_amd/lib/libswz_v*.so)
cd $GRAFT_REPO_ROOT
for lib in schwarzwald_amd/lib/libswz_gpu.so schwarzwald_amd/lib/libswz_v*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib $VAR_ENV"
  env $VAR_ENV SWZ_GPU_LIBRARY=$PWD/$lib SWZ_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --md-mode ${MD_MODE:-exact} --also "" 2>&1 >/tmp/out.json | grep -E "block path|thread 0" | tail -2 | cut -c1-60,150-500
  grep -o '"ms_per_step": [0-9.]*' /tmp/out.json | head -2
done
