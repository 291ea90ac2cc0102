# runs a pytest selection with every libswz_v*.so variant (SWZ_GPU_LIBRARY), then times them (probe.sh variants)
cd $GRAFT_REPO_ROOT
for lib in schwarzwald_amd/lib/libswz_v*.so; do
  [ -f "$lib" ] || continue
  echo "== tests with $lib"
  SWZ_GPU_LIBRARY=$PWD/$lib timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_min_distance_keys.py} -x -q -m gpu -k "${KEXPR:-min_distance or sparse}" 2>&1 | tail -2
done
bash tools/probe.sh variants --md-mode exact
